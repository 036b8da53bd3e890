"""The LDS layout of a bench config's model (mjh_debug_lds_layout): arrays in address order with their sizes, the granule count (1280 B) and
the workgroups per CU that follow — which array costs a granule.     python tools/lds_layout.py [s24|s24d|c2|c3|c4|c5]     (no GPU needed for
scene models; c4 / c5 load fixtures)"""
import sys, os, types, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mujoco_sim_amd as ms


def layout(model):
    lib = ms.capi.load()
    buf = C.create_string_buffer(8192)
    n = lib.mjh_debug_lds_layout(model.ptr, buf, 8192)
    assert n > 0
    kv = [ln.split() for ln in buf.value.decode().strip().splitlines()]
    return [(k, int(v)) for k, v in kv]


def show(model, label):
    kv = layout(model)
    tail = dict(kv[kv.index(next(x for x in kv if x[0] == "total")):])
    arrs = [(k, v) for k, v in kv if k not in tail]
    lds = sorted([(v, k) for k, v in arrs if v >= 0])
    print(f"== {label}: total {tail['total']} floats, lds_bytes {tail['lds_bytes']} ({-(-tail['lds_bytes'] // 1280)} granules -> {128 // -(-tail['lds_bytes'] // 1280)} per CU), "
          f"assemble-only {tail['lds_bytes_pre']} B, k1 {tail['k1_floats']}, maxcon {tail['maxcon']}, maxblk {tail['maxblk']}, rowW {tail['rowW']}, nstage {tail['nstage']}, big {tail['big']}")
    offs = sorted(set(v for v, _ in lds)) + [tail["total"]]
    for o, nxt in zip(offs[:-1], offs[1:]):
        names = [k for v, k in lds if v == o]
        print(f"  {o:6d} .. {nxt:6d}  ({nxt - o:5d} floats)  {' = '.join(names)}")
    g = [k for k, v in arrs if v < 0]
    if g:
        print("  global slice:", " ".join(g))


if __name__ == "__main__":
    which = sys.argv[1:] or ["s24", "s24d", "c3"]
    for w in which:
        if w == "s24": show(ms.scene("s24"), "s24")
        elif w == "s24d": show(ms.scene("s24pen", 0.175, 96), "s24d (capacity 96)")
        elif w == "c3":
            m = ms.scene("arm7", 1); show(m, "arm7"); show(m.replicate(4), "c3: arm7 x 4 per wavefront")
        elif w == "c2":
            m = ms.scene("boxpile", 64); m.c.maxcon = 600; m.c.maxefc = 2400; show(m, "c2")
        else:
            print("unknown", w)
