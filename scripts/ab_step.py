"""Same-process, same-box A/B of engine options on the full optimisation step (boxes of the pool differ by +-3 %, so variants are
timed back to back on one model, twice, in alternating order).

    python scripts/ab_step.py [--microbatch 256] [--steps 3] name:key=value,key=value ...
e.g. python scripts/ab_step.py --microbatch 256 r2:ksplit_min_items=128,group_adaln=0,use_arena=0 new:ksplit_min_items=192
Keys are DiTEngine attributes (ints / bools), or env.NAME for a library switch read with getenv (MD_GEMM_NO_*)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--microbatch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--stage", default="res_256_pretrain")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    st = bench.Stage(a.stage, "MicroDiT_XL_2", 2048, a.microbatch, 1, 0)
    eng = st.model.dit.engine
    base = {}
    res = {}
    for rnd in range(a.rounds):
        for v in (a.variants if rnd % 2 == 0 else a.variants[::-1]):
            name, _, opts = v.partition(":")
            kv = dict(o.split("=") for o in opts.split(",") if o)
            for k in base:
                setattr(eng, k, base[k])
            for k in [k for k in os.environ if k.startswith("MD_GEMM_")]:
                del os.environ[k]
            for k, val in kv.items():
                if k.startswith("env."):                # library switches read with getenv at every launch (e.g. env.MD_GEMM_NO_W4=1)
                    os.environ[k[4:]] = val
                    continue
                base.setdefault(k, getattr(eng, k))
                setattr(eng, k, type(getattr(eng, k))(int(val)))
            for ar in (eng._tape_arena, eng._scratch_arena):     # the launch sequence may change with the options: re-measure
                ar.peaks.clear()
                ar.buf = None
            e, loss = st.timed(a.steps, 1, 1)
            res.setdefault(name, []).append(2048 * a.steps / e)
            print(f"{name:24s} round {rnd}: {2048 * a.steps / e:8.1f} img/s  loss {loss:.5f}", flush=True)
    print(json.dumps({"microbatch": a.microbatch, "stage": a.stage, "img_per_s": res}))


if __name__ == "__main__":
    main()
