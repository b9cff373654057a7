"""Per-robot state record: layout access and default initialisation (host side, numpy).

The record layout is defined once, in include/openroborl_hip.h (ORR_STATE_FIELDS); the C-ABI
library reports it through orr_layout_*().  The record is a row of ORR_STATE_STRIDE 32-bit words:
float32 fields and int32 fields (bit-cast in place).
"""
import numpy as np

from . import _abi


class Layout(object):
    """name -> (offset, size, is_int), queried from a library exposing <prefix>_layout_*()."""

    def __init__(self, lib, prefix="orr"):
        import ctypes as C
        g = lambda n: getattr(lib, "%s_%s" % (prefix, n))
        g("layout_name").restype = C.c_char_p
        g("layout_name").argtypes = [C.c_int32]
        for fn in ("layout_offset", "layout_size", "layout_is_int"):
            g(fn).restype = C.c_int32
            g(fn).argtypes = [C.c_int32]
        g("layout_count").restype = C.c_int32
        g("state_stride").restype = C.c_int32
        self.stride = int(g("state_stride")())
        self.fields = {}
        self.order = []
        for i in range(int(g("layout_count")())):
            name = g("layout_name")(i).decode()
            self.fields[name] = (int(g("layout_offset")(i)), int(g("layout_size")(i)), bool(g("layout_is_int")(i)))
            self.order.append(name)
        assert self.stride == _abi.STATE_STRIDE

    def sl(self, name):
        off, size, _ = self.fields[name]
        return slice(off, off + size)

    def is_int(self, name):
        return self.fields[name][2]

    def int_mask(self):
        m = np.zeros(self.stride, dtype=bool)
        for name, (off, size, is_int) in self.fields.items():
            if is_int:
                m[off:off + size] = True
        return m


def grid_offset(robot_index):
    """minitaur.py:246-248: x -= 2*(i//4), y += 2*(i%4)."""
    i = np.asarray(robot_index)
    return np.stack([-2.0 * (i // 4), 2.0 * (i % 4)], axis=-1).astype(np.float64)


def default_state(layout, n, models, robot_type, clip_id, robot_index, legacy_grid=False, ctrl_latency=0.002,
                  max_ep_steps=600, base_damping=(0.0, 0.0)):
    """Fresh [n, stride] float32 state (ints bit-cast) holding only what reset() does not write:
    identity randomisation parameters, robot type / clip / index, grid slot.

    models: list of model dicts indexed by robot type; robot_type / clip_id / robot_index: int arrays [n].
    """
    st = np.zeros((n, layout.stride), dtype=np.float32)
    iv = st.view(np.int32)
    robot_type = np.broadcast_to(np.asarray(robot_type, dtype=np.int32), (n,))
    clip_id = np.broadcast_to(np.asarray(clip_id, dtype=np.int32), (n,))
    robot_index = np.broadcast_to(np.asarray(robot_index, dtype=np.int32), (n,))
    st[:, layout.sl("STRENGTH")] = 1.0
    st[:, layout.sl("LATENCY")] = ctrl_latency            # laikago.py:27 CTRL_LATENCY
    st[:, layout.sl("MASS_RATIO")] = 1.0
    st[:, layout.sl("INERTIA_RATIO")] = 1.0
    st[:, layout.sl("BASE_DAMPING")] = np.asarray(base_damping, dtype=np.float32)
    st[:, layout.sl("ORIGIN_ROT")] = np.array([0, 0, 0, 1], dtype=np.float32)
    st[:, layout.sl("QUAT")] = np.array([0, 0, 0, 1], dtype=np.float32)
    for t, m in enumerate(models):
        sel = robot_type == t
        if m is not None and sel.any():
            st[sel, layout.sl("FOOT_MU")] = m["foot_friction"]
    if legacy_grid:
        st[:, layout.sl("GRID_OFFSET")] = grid_offset(robot_index).astype(np.float32)
    iv[:, layout.sl("ROBOT_TYPE")] = robot_type[:, None]
    iv[:, layout.sl("CLIP_ID")] = clip_id[:, None]
    iv[:, layout.sl("ROBOT_INDEX")] = robot_index[:, None]
    iv[:, layout.sl("MAX_EP_STEPS")] = max_ep_steps
    iv[:, layout.sl("RING_HEAD")] = _abi.RING_DEPTH - 1
    return st


def to_float64(layout, st32):
    """float32 record (ints bit-cast) -> float64 record with ints stored as values (oracle format)."""
    st32 = np.ascontiguousarray(st32, dtype=np.float32)
    out = st32.astype(np.float64)
    m = layout.int_mask()
    out[:, m] = st32.view(np.int32)[:, m].astype(np.float64)
    return out


def from_float64(layout, st64):
    out = st64.astype(np.float32)
    m = layout.int_mask()
    out.view(np.int32)[:, m] = np.rint(st64[:, m]).astype(np.int32)
    return out
