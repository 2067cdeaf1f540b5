"""RCCL on the one GPU of the box: the multi-GPU helpers of semstereo_amd.dist -- process-group set-up
over the `nccl` backend (= RCCL on ROCm), the one-time weight broadcast of a HotSegment, the padded all_gather of device
disparities, the float64 metric all_reduce, teardown -- run in a FRESH process with a world of ONE rank.  The N > 1
curve itself is the driver's (8-GPU node); what is pinned here is that every RCCL call of the path executes on an MI355X
(communicator creation, device buffers, collectives on the current stream) and returns what gloo returns on CPU
(tests/test_dist_gloo.py).  Also: `bench.py` under SS_DIST_FORCE_INIT=1 runs its whole step loop over that group."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, os.environ["SS_ROOT"])
import torch, torch.distributed as dist
import semstereo_amd
from semstereo_amd import dist as sd
rank, world, local = sd.init_from_env("nccl", force=True)
assert (rank, world, local) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
out = {"rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "backend": dist.get_backend()}
seg = semstereo_amd.HotSegment(64).to(dev).eval()
before = [p.detach().clone() for p in seg.parameters()]
sd.broadcast_module(seg, src=0)                                   # one ncclBroadcast per parameter / buffer
torch.cuda.synchronize()
out["broadcast_tensors"] = len(list(seg.parameters())) + len(list(seg.buffers()))
out["broadcast_identity"] = all(torch.equal(a, b) for a, b in zip(before, seg.parameters()))
g = torch.Generator(device=dev).manual_seed(3)
local_disp = torch.randn(3, 40, 56, generator=g, device=dev)      # this rank's block of [b,H,W] disparities (odd sizes)
full = sd.gather_batch(local_disp, 3)                             # padded all_gather of DEVICE tensors
out["gather_ok"] = bool(full.is_cuda and torch.equal(full, local_disp))
pairs, err, pix, tmax = sd.reduce_metrics(3, 0.125, local_disp.numel(), 0.75, dev)     # float64 device all_reduce SUM / MAX
out["reduce"] = [pairs, err, pix, tmax]
# the hot segment itself between two barriers, as bench.py brackets it
fl4 = torch.randn(1, 128, 32, 32, generator=g, device=dev); fr4 = torch.randn(1, 128, 32, 32, generator=g, device=dev)
fl8 = torch.randn(1, 256, 16, 16, generator=g, device=dev); fr8 = torch.randn(1, 256, 16, 16, generator=g, device=dev)
dist.barrier()
with torch.no_grad():
    r = seg(fl4, fr4, fl8, fr8)
torch.cuda.synchronize()
dist.barrier()
out["segment_finite"] = bool(torch.isfinite(r["pred"]).all())
dist.destroy_process_group()
out["destroyed"] = not dist.is_initialized()
print("RESULT " + json.dumps(out), flush=True)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               NCCL_DEBUG="VERSION", SS_ROOT=ROOT)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_world_of_one_runs_every_collective_of_the_path():
    p = subprocess.run([sys.executable, "-c", _SCRIPT], env=_env(), capture_output=True, text=True, timeout=600)
    log = p.stdout + p.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "nccl_world1.log"), "w") as f:
        f.write(log)
    assert p.returncode == 0, log[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, log[-4000:]
    out = json.loads(line[-1][len("RESULT "):])
    print("RCCL version", out["rccl_version"])
    assert out["backend"] == "nccl" and out["rccl_version"]
    assert out["broadcast_identity"] and out["broadcast_tensors"] > 100
    assert out["gather_ok"] and out["reduce"] == [3.0, 0.125, 3.0 * 40 * 56, 0.75]
    assert out["segment_finite"] and out["destroyed"]
    # NCCL_DEBUG=VERSION makes the library announce itself: the communicator really was RCCL's
    assert any("NCCL version" in ln or "RCCL version" in ln for ln in log.splitlines()), log[-2000:]


def test_bench_step_loop_over_a_one_rank_rccl_group():
    """bench.py's own control flow (init, broadcast, barriers, timed loop, reduce_metrics, destroy) over RCCL at N = 1."""
    env = _env()
    env["SS_DIST_FORCE_INIT"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--height", "256",
                        "--width", "256", "--maxdisp", "64", "--no-other-engines", "--no-cpu-baseline", "--no-kernel-timers"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["value"] > 0
    # the group as torch.distributed reports it, this rank's own rate and the all_gather of the disparities that follows the forward
    d = res["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and len(d["per_rank_pairs_per_s"]) == 1
    assert d["all_gather_payload_bytes"] == 1 * 1 * 64 * 64 * 4 and d["all_gather_shape"] == [1, 1, 64, 64]
