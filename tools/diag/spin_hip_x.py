"""POST-HOC diagnostic, after the cross-validation's hold-outs had been seen - nothing was chosen from it: why is
`laikago_spin` never predicted by a table that was not fitted on it?  CPU oracle, 16 robots x 300 steps, the spin policy on the chosen tables
of the six splits (profiles/r06_laikago_cv.json), with the toe friction or the hips' lengthwise position alone changed.  Reading: friction is
not it; hip_x is - the tables fitted without spin sit at hip_x = 0.27 (an edge of the interval the trots weakly prefer) and turn 15-25 % too
slowly, every robot ends by the root-rotation test; with hip_x alone put to 0.19-0.23 the held-out spin policy finishes 44-94 % on them, and
the spin-fitted table of split 3 with hip_x = 0.27 loses everybody.  The yaw rate of a spin is foot speed over the hips' radius: spin is
the one policy that identifies the wheelbase, and it says 0.19-0.23 (laikago.py:54-59 says 0.21).
usage: python tools/diag/spin_hip_x.py > profiles/r06_spin_hip_x.txt"""
import sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import identify_r6 as I
from openroborl_amd import config, motion, robots
from tests import oracle_lib as ol
def run(model, n=16, steps=300, label=""):
    W=np.load(os.path.join(ol.GOLDEN,"policy_laikago_spin.npz")); w={k:W[k].astype(np.float64) for k in W.files}
    clip=motion.MotionClip("laikago_spin")
    cfg=config.make_config(n, sim_params=config.load_sim_params(None), mode="test", enable_randomizer=False, seed=1, num_procs=1, auto_reset=False)
    orc=ol.OracleEnv(cfg,[model,None,None,None],[clip],n,robot_type=np.zeros(n,dtype=np.int32),clip_id=np.zeros(n,dtype=np.int32),threads=8)
    orc.field("FOOT_MU")[:]=model["foot_friction"]; obs=orc.reset(); orc.field("FOOT_MU")[:]=model["foot_friction"]
    alive=np.ones(n,bool); wz=[]; wref=[]; length=np.zeros(n); reasons=np.zeros(n,int); rew_s=0; cnt=0
    for s in range(steps):
        h=np.maximum(obs@w["model__pi_fc0__w_0"]+w["model__pi_fc0__b_0"],0); h=np.maximum(h@w["model__pi_fc1__w_0"]+w["model__pi_fc1__b_0"],0)
        a=np.clip(h@w["model__pi__w_0"]+w["model__pi__b_0"],-2*np.pi,2*np.pi)
        obs,rew,done=orc.step(a)
        length+=alive
        r=orc.field("DONE_REASON")[:,0].astype(int); failed=done&((r&~8)!=0); reasons=np.where(alive&failed,r,reasons); alive&=~failed
        if alive.any():
            wz.append(orc.field("ANGVEL")[alive,2].mean()); wref.append(orc.field("REF_VEL")[alive,5].mean()); rew_s+=rew[alive].mean(); cnt+=1
    orc.close()
    print("%-46s mu %.2f: up %.2f len %5.1f  mean yaw rate %+.3f rad/s  r/step %.3f  rot-fail %d fall %d"%(label, model["foot_friction"], alive.mean(), length.mean(), np.mean(wz), rew_s/max(cnt,1), ((reasons&4)!=0).sum(), ((reasons&1)!=0).sum()))
recs={r['split']:r for r in json.load(open(os.path.join(ROOT,'profiles','r06_laikago_cv.json')))['splits']}
run(robots.laikago(), label="shipped (all four in the fit)")
for i in (5,1,2):
    th=recs[i]['chosen']['theta']; run(I.build_model("laikago",th), label="split %d table (fit %s; spin held out)"%(i,'+'.join(p.replace('laikago_','') for p in recs[i]['fit'])))
    for mu in (0.5,0.3):
        run(I.build_model("laikago",dict(th,foot_friction=mu)), label="   the same with toe friction %.1f"%mu)
for i in (3,0):
    th=recs[i]['chosen']['theta']; run(I.build_model("laikago",th), label="split %d table (fit %s)"%(i,'+'.join(p.replace('laikago_','') for p in recs[i]['fit'])))
    run(I.build_model("laikago",dict(th,foot_friction=1.2)), label="   the same with toe friction 1.2")
print("---- hip_x alone")
for i in (5,1,2):
    th=recs[i]['chosen']['theta']
    for hx in (0.192, 0.21, 0.23):
        run(I.build_model("laikago",dict(th,hip_x=hx)), label="split %d table with hip_x %.3f (was %.3f)"%(i,hx,th['hip_x']))
th=recs[3]['chosen']['theta']
for hx in (0.23,0.27):
    run(I.build_model("laikago",dict(th,hip_x=hx)), label="split 3 table with hip_x %.3f (was %.3f)"%(hx,th['hip_x']))
