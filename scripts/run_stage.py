"""Run a few full optimisation steps of one reference stage (bench.STAGES) for profiling:  python scripts/run_stage.py res_512_pretrain [steps] [microbatch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "res_512_pretrain"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mb = int(sys.argv[3]) if len(sys.argv) > 3 else bench.STAGES[name]["microbatch"]
torch.cuda.set_device(0)
st = bench.Stage(name, "MicroDiT_XL_2", 2048, mb, 1, 0)
e, loss = st.timed(steps, 1, 1)
print(f"{name}: {2048 * steps / e:.1f} images/s, {e / steps * 1e3:.1f} ms per step, microbatch {st.microbatch}, loss {loss:.4f}")
