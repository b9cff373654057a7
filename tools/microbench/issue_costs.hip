// Issue cost of instruction kinds for ONE wave per SIMD on gfx950 (development aid; DESIGN.md section 6 quotes the results).
// Every test is 256 repetitions of a short pattern between two s_memtime reads, run by a single wave; printed: cycles per repetition.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_costs tools/microbench/issue_costs.hip && /tmp/issue_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1(id) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); if (threadIdx.x == 0) out[id] = t1 - t0

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k(long long* out, float* sink) {
  __shared__ float lds[1024];
  long long t0, t1;
  float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f, kf = 0.5f;
  v2f p = {1.0f, 2.0f};
  v4f q = {0, 0, 0, 0};
  int s0 = 1;
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0.0f;
  __syncthreads();
  int addr = (int)(size_t)lds + ((threadIdx.x * 4) & 1023);
  T0(); T1(0);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %1, %0\n\t") : "+v"(a) : "v"(kf)); T1(1);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\t") : "+v"(a), "+v"(b) : "v"(kf)); T1(2);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %1, %0\n\ts_nop 0\n\t") : "+v"(a) : "v"(kf)); T1(3);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %1, %0\n\ts_nop 1\n\t") : "+v"(a) : "v"(kf)); T1(4);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %0\n\ts_add_u32 %1, %1, 1\n\t") : "+v"(a), "+s"(s0) : "v"(kf) : "scc"); T1(5);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %0\n\ts_nop 1\n\tv_mov_b32_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a), "+v"(b) : "v"(kf)); T1(6);
  T0(); asm volatile(REP256("v_pk_fma_f32 %0, %0, %0, %0\n\t") : "+v"(p)); T1(7);
  {
    int z;
    T0(); asm volatile(REP256("ds_read_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(addr), "=&v"(z) : : "memory"); T1(8);
    T0(); asm volatile(REP256("ds_read_b128 v[100:103], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v100\n\t") : "+v"(addr) : : "memory", "v100", "v101", "v102", "v103"); T1(9);
  }
  T0(); asm volatile(REP256("v_rcp_f32 %0, %0\n\t") : "+v"(a)); T1(10);
  T0(); asm volatile(REP256("v_rcp_f32 %0, %1\n\tv_fma_f32 %2, %2, %3, %2\n\t") : "=&v"(c), "+v"(b), "+v"(d) : "v"(kf)); T1(11);
  T0(); asm volatile(REP256("v_readlane_b32 %0, %1, 3\n\t") : "=s"(s0) : "v"(a)); T1(12);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %0\n\tv_readlane_b32 %1, %3, 3\n\t") : "+v"(a), "=&s"(s0) : "v"(kf), "v"(b)); T1(13);
  T0(); asm volatile(REP256("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0\n\t") : "+v"(a) : : "a0"); T1(14);
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(a) : "v"(kf)); T1(15);
  T0(); asm volatile(REP256("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(c) : "v"(a), "v"(b) : "vcc"); T1(16);
  T0(); asm volatile(REP256("v_cmp_lt_f32 vcc, %1, %2\n\ts_and_saveexec_b64 s[10:11], vcc\n\ts_cbranch_execz 1f\n\tv_fma_f32 %0, %0, %3, %0\n1:\n\ts_or_b64 exec, exec, s[10:11]\n\t")
                     : "+v"(c) : "v"(a), "v"(b), "v"(kf) : "vcc", "s10", "s11"); T1(17);
  T0(); asm volatile(REP256("ds_write_b32 %0, %1\n\t") : : "v"(addr), "v"(a) : "memory"); T1(18);
  {
    float z;
    T0(); asm volatile(REP256("ds_read_b32 %1, %2\n\tv_fma_f32 %0, %0, %3, %0\n\tv_fma_f32 %0, %0, %3, %0\n\tv_fma_f32 %0, %0, %3, %0\n\tv_fma_f32 %0, %0, %3, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\t")
                       : "+v"(a), "=&v"(z) : "v"(addr), "v"(kf) : "memory"); T1(19);
    T0(); asm volatile(REP256("ds_read_b32 %1, %2\n\t" REP16("v_fma_f32 %0, %0, %3, %0\n\t") "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\t")
                       : "+v"(a), "=&v"(z) : "v"(addr), "v"(kf) : "memory"); T1(20);
  }
  T0(); asm volatile(REP256("v_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %1, %1, %2, %1\n\tv_mov_b32_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a), "+v"(b) : "v"(kf)); T1(21);
  T0(); asm volatile(REP256("s_add_u32 %0, %0, 1\n\t") : "+s"(s0) : : "scc"); T1(22);
  T0(); asm volatile(REP256("v_mov_b32_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a), "+v"(b)); T1(23);  // illegal back-to-back (hazard) -- timing only
  T0(); asm volatile(REP256("v_sin_f32 %0, %0\n\t") : "+v"(a)); T1(24);
  T0(); asm volatile(REP256("v_med3_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c)); T1(25);
  {
    // exec-mask and branch patterns; s[12:13] = lanes 0..11 of every 16, s[14:15] = 0
    asm volatile("s_mov_b64 s[12:13], 0x0fff0fff\n\ts_mov_b32 s13, 0x0fff0fff\n\ts_mov_b64 s[14:15], 0" ::: "s12", "s13", "s14", "s15");
    T0(); asm volatile(REP256("s_and_saveexec_b64 s[10:11], s[12:13]\n\tv_fma_f32 %0, %0, %1, %0\n\ts_or_b64 exec, exec, s[10:11]\n\t") : "+v"(a) : "v"(kf) : "s10", "s11", "scc"); T1(26);
    T0(); asm volatile(REP256("s_and_saveexec_b64 s[10:11], s[12:13]\n\ts_cbranch_execz 1f\n\tv_fma_f32 %0, %0, %1, %0\n1:\n\ts_or_b64 exec, exec, s[10:11]\n\t") : "+v"(a) : "v"(kf) : "s10", "s11", "scc"); T1(27);
    T0(); asm volatile(REP256("v_cmp_lt_f32 vcc, %1, %2\n\ts_and_saveexec_b64 s[10:11], vcc\n\tv_fma_f32 %0, %0, %3, %0\n\ts_or_b64 exec, exec, s[10:11]\n\t") : "+v"(c) : "v"(a), "v"(b), "v"(kf) : "vcc", "s10", "s11", "scc"); T1(28);
    T0(); asm volatile(REP256("s_mov_b64 exec, s[12:13]\n\tv_fma_f32 %0, %0, %1, %0\n\ts_mov_b64 exec, -1\n\t") : "+v"(a) : "v"(kf)); T1(29);
    T0(); asm volatile(REP256("s_and_saveexec_b64 s[10:11], s[14:15]\n\ts_cbranch_execz 1f\n\tv_fma_f32 %0, %0, %1, %0\n1:\n\ts_or_b64 exec, exec, s[10:11]\n\t") : "+v"(a) : "v"(kf) : "s10", "s11", "scc"); T1(30);
    T0(); asm volatile(REP256("s_cmp_eq_u32 %1, 0x7fff\n\ts_cbranch_scc1 1f\n\tv_fma_f32 %0, %0, %2, %0\n1:\n\t") : "+v"(a) : "s"(s0), "v"(kf) : "scc"); T1(31);
    T0(); asm volatile(REP256("s_cmp_lg_u32 %1, 0x7fff\n\ts_cbranch_scc1 1f\n\tv_fma_f32 %0, %0, %2, %0\n1:\n\tv_fma_f32 %0, %0, %2, %0\n\t") : "+v"(a) : "s"(s0), "v"(kf) : "scc"); T1(32);
    T0(); asm volatile(REP256("v_cndmask_b32 %0, %0, %1, s[12:13]\n\t") : "+v"(a) : "v"(b)); T1(33);
    T0(); asm volatile(REP256("ds_read2_b32 v[100:101], %0 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v100\n\t") : "+v"(addr) : : "memory", "v100", "v101"); T1(34);
    T0(); asm volatile(REP256("ds_read_b64 v[100:101], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v100\n\t") : "+v"(addr) : : "memory", "v100", "v101"); T1(35);
    T0(); asm volatile(REP256("ds_write_b32 %0, %1\n\tds_read_b32 v100, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, v100\n\t") : "+v"(addr) : "v"(0.0f) : "memory", "v100"); T1(36);
    T0(); asm volatile(REP256("ds_bpermute_b32 v100, %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %1, %1, v100\n\t") : "+v"(addr), "+v"(a) : : "memory", "v100"); T1(37);
  }
  {
    v2f p1 = {1.5f, 2.5f}, p2 = {0.5f, 0.25f}, p3 = {3.0f, 4.0f};
    T0(); asm volatile(REP256("v_pk_fma_f32 %0, %1, %2, %0\n\t") : "+v"(p) : "v"(p1), "v"(p2)); T1(38);                       // 3 distinct 64-bit sources
    T0(); asm volatile(REP256("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1\n\t") : "+v"(p), "+v"(p3) : "v"(p1), "v"(p2)); T1(39);   // two independent
    T0(); asm volatile(REP256("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t") : "+v"(p) : "v"(p1), "v"(p2)); T1(40);     // scalar splat
    T0(); asm volatile(REP256("v_pk_mul_f32 %0, %1, %0\n\t") : "+v"(p) : "v"(p1)); T1(41);
    T0(); asm volatile(REP256("v_pk_add_f32 %0, %1, %0\n\t") : "+v"(p) : "v"(p1)); T1(42);
    T0(); asm volatile(REP256("v_fma_f32 %0, %1, %2, %0\n\t") : "+v"(a) : "v"(b), "v"(c)); T1(43);                               // scalar, 3 distinct sources
    T0(); asm volatile(REP256("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t") : "+v"(p) : "v"(p1), "v"(p2)); T1(44);
    sink[threadIdx.x + 64] = p1.x + p2.x + p3.x;
    // LDS issue cost by width: independent operations (no wait inside the block), 4 VALU between them
    T0(); asm volatile(REP256("ds_read_b32 v100, %0\n\t") : : "v"(addr) : "memory", "v100"); T1(45);
    T0(); asm volatile(REP256("ds_read_b64 v[100:101], %0\n\t") : : "v"(addr & ~7) : "memory", "v100", "v101"); T1(46);
    T0(); asm volatile(REP256("ds_read_b128 v[100:103], %0\n\t") : : "v"(addr & ~15) : "memory", "v100", "v101", "v102", "v103"); T1(47);
    T0(); asm volatile(REP256("ds_read2_b32 v[100:101], %0 offset1:8\n\t") : : "v"(addr) : "memory", "v100", "v101"); T1(48);
    T0(); asm volatile(REP256("ds_write_b64 %0, %1\n\t") : : "v"(addr & ~7), "v"(p1) : "memory"); T1(49);
    T0(); asm volatile(REP256("ds_write_b128 %0, %1\n\t") : : "v"(addr & ~15), "v"(q) : "memory"); T1(50);
    T0(); asm volatile(REP256("ds_read_b128 v[100:103], %1\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %0, %0, %2, %0\n\t") : "+v"(a) : "v"(addr & ~15), "v"(kf) : "memory", "v100", "v101", "v102", "v103"); T1(51);
    T0(); asm volatile(REP256("ds_read_b32 v100, %1\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %0, %0, %2, %0\n\tv_fma_f32 %0, %0, %2, %0\n\t") : "+v"(a) : "v"(addr), "v"(kf) : "memory", "v100"); T1(52);
  }
  {
    // VGPR banks (register number mod 4): three sources from one bank vs from three banks; explicit registers
    asm volatile("v_mov_b32 v100, 1.0\n\tv_mov_b32 v104, 0.5\n\tv_mov_b32 v108, 0.25\n\tv_mov_b32 v101, 0.5\n\tv_mov_b32 v102, 0.25\n\tv_mov_b32 v103, 2.0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0"
                 ::: "v100", "v101", "v102", "v103", "v104", "v108", "v112", "v113");
    T0(); asm volatile(REP256("v_fma_f32 v112, v100, v104, v108\n\t") ::: "v112"); T1(53);                  // sources in banks 0,0,0 (independent)
    T0(); asm volatile(REP256("v_fma_f32 v112, v100, v101, v102\n\t") ::: "v112"); T1(54);                  // sources in banks 0,1,2
    T0(); asm volatile(REP256("v_fma_f32 v112, v100, v104, v112\n\t") ::: "v112"); T1(55);                  // dependent, banks 0,0,0
    T0(); asm volatile(REP256("v_fma_f32 v113, v100, v102, v113\n\t") ::: "v113"); T1(56);                  // dependent, banks 0,2,1
    T0(); asm volatile(REP256("v_fmac_f32 v112, v100, v104\n\t") ::: "v112"); T1(57);                       // fmac dependent, banks 0,0,(0)
    T0(); asm volatile(REP256("v_fmac_f32 v113, v100, v102\n\t") ::: "v113"); T1(58);                       // fmac dependent, banks 0,2,(1)
    T0(); asm volatile(REP256("v_fmac_f32_dpp v113, v100, v102 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") ::: "v113"); T1(59);
    T0(); asm volatile(REP256("v_fmac_f32_dpp v112, v100, v104 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") ::: "v112"); T1(60);
    T0(); asm volatile(REP256("v_med3_f32 v113, v100, v104, v108\n\t") ::: "v113"); T1(61);
    T0(); asm volatile(REP256("v_med3_f32 v113, v100, v101, v102\n\t") ::: "v113"); T1(62);
  }
  sink[threadIdx.x] = a + b + c + d + (float)addr + (float)s0 + p.x + q.x;
}

int main() {
  long long* out; float* sink;
  (void)hipMalloc(&out, 64 * sizeof(long long)); (void)hipMalloc(&sink, 128 * sizeof(float));
  (void)hipMemset(out, 0, 64 * sizeof(long long));
  const char* names[] = {"empty (timer overhead)", "v_fma dependent chain", "2 independent v_fma (per pair)", "v_fma + s_nop 0", "v_fma + s_nop 1",
                         "v_fma + s_add", "v_fma + s_nop 1 + v_mov_dpp", "v_pk_fma_f32 dependent", "ds_read_b32 dependent round trip (+waitcnt +v_add)",
                         "ds_read_b128 dependent round trip", "v_rcp dependent", "v_rcp + independent v_fma", "v_readlane", "v_fma + v_readlane",
                         "accvgpr write + read", "v_fma + satisfied s_waitcnt", "v_cmp + v_cndmask", "divergent if (cmp, saveexec, branch, fma, or)",
                         "ds_write_b32", "ds_read + 4 v_fma + wait + add", "ds_read + 16 v_fma + wait + add", "2 v_fma + v_mov_dpp (1 filler)",
                         "s_add dependent", "2 v_mov_dpp back to back", "v_sin dependent", "v_med3 dependent",
                         "saveexec(sgpr mask) + v_fma + s_or exec", "saveexec + cbranch_execz (not taken) + v_fma + s_or", "v_cmp + saveexec(vcc) + v_fma + s_or",
                         "s_mov exec + v_fma + s_mov exec", "saveexec + cbranch_execz (TAKEN) + s_or", "s_cmp + s_cbranch_scc (not taken) + v_fma",
                         "s_cmp + s_cbranch_scc (TAKEN over 1) + v_fma", "v_cndmask with sgpr mask", "ds_read2_b32 round trip", "ds_read_b64 round trip",
                         "ds_write + ds_read same address round trip", "ds_bpermute round trip",
                         "v_pk_fma_f32, 3 distinct sources (dependent)", "2 independent v_pk_fma_f32 (per pair)", "v_pk_fma_f32 with op_sel splat",
                         "v_pk_mul_f32 dependent", "v_pk_add_f32 dependent", "v_fma_f32, 3 distinct sources", "v_pk_fma_f32 with op_sel swap + neg",
                         "ds_read_b32 back to back (issue)", "ds_read_b64 back to back", "ds_read_b128 back to back", "ds_read2_b32 back to back",
                         "ds_write_b64 back to back", "ds_write_b128 back to back", "ds_read_b128 + 4 v_fma (no wait)", "ds_read_b32 + 4 v_fma (no wait)",
                         "v_fma, sources in VGPR banks 0,0,0", "v_fma, sources in banks 0,1,2", "v_fma dependent, banks 0,0,0", "v_fma dependent, banks 0,2,1",
                         "v_fmac dependent, banks 0,0,0", "v_fmac dependent, banks 0,2,1", "v_fmac_dpp dependent, banks 0,2,1", "v_fmac_dpp dependent, banks 0,0,0",
                         "v_med3, banks 0,0,0", "v_med3, banks 0,1,2"};
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink);
    (void)hipDeviceSynchronize();
  }
  long long h[64];
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 63; i++) printf("%-52s %8.2f cycles per repetition\n", names[i], (double)(h[i] - h[0]) / 256.0);
  return 0;
}
