"""`python bench.py --gpus N` must be runnable as typed: with WORLD_SIZE unset the parent spawns
its own N ranks without touching the GPU, relays rank 0's JSON line and fails if a rank fails.  Rehearsed here on
CPU with `--dry-launch` (gloo, stand-in step): launcher, rendezvous on 127.0.0.1, barriers around the timed region,
SUM / MAX metric reduction, rank != 0 teardown.  The counterpart of nn.DataParallel in test_us3d.py:58."""
import json
import os
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(line) for line in text.splitlines() if line.startswith("{")]


def test_self_launch_two_ranks_prints_one_json_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "4", "--warmup", "1", "--batch", "4"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    res = lines[0]
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["warmup"] == 1 and res["dry_launch"] is True
    assert res["pairs_counted"] == 2 * 4 * 4            # SUM over both ranks of batch x steps
    assert res["ms_per_step"] > 0
    # what makes a SCALE run self-verifying: the group's size as torch.distributed reports it, every rank's own
    # rate, and the bytes of the one collective that follows the forward
    d = res["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo"
    assert len(d["per_rank_pairs_per_s"]) == 2 and all(r > 0 for r in d["per_rank_pairs_per_s"])
    assert d["all_gather_payload_bytes"] > 0 and d["all_gather_shape"][0] == 2 * 4
    assert res["steady_state"]["steps"] >= 4 and res["steady_state"]["pairs_per_s"] > 0


import pytest  # noqa: E402


@pytest.mark.parametrize("argv, config, batch", [
    (["--batch", "4"], "configs[3]", 4),                                                   # 1024^2 / 128, batch 32 over 8 GPUs
    (["--height", "2048", "--width", "2048", "--maxdisp", "192"], "configs[4]", 1),        # 2048^2 / 192, batch 8 over 8 GPUs
])
def test_eight_rank_rehearsal_of_the_sharded_configs(argv, config, batch):
    """the two 8-GPU configurations of BASELINE.json as `--dry-launch` rehearsals -- eight self-launched ranks, rank
    r bound to device r, per-rank workload as the config says, the padded all_gather of the whole batch, the reductions --
    so that the launch line of a SCALE run is known to hold before an 8-GPU node appears (reference: nn.DataParallel over
    the visible GPUs, test_us3d.py:58).  No scaling curve exists; this asserts control flow only."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch", "--steps", "2", "--warmup", "1"] + argv,
                       env=_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    res = lines[0]
    assert res["n_gpus"] == 8 and res["pairs_counted"] == 8 * batch * 2
    d = res["dist"]
    assert d["world_size"] == 8 and len(d["per_rank_pairs_per_s"]) == 8
    assert sorted(b["rank"] for b in d["rank_devices"]) == list(range(8))
    assert all(b["local_rank"] == b["rank"] and b["device"].endswith("cuda:%d)" % b["rank"]) for b in d["rank_devices"])
    assert d["all_gather_shape"][0] == 8 * batch
    assert d["workload_per_rank"]["config"] == config and d["workload_per_rank"]["pairs_per_step"] == batch


def test_self_launch_fails_when_a_rank_fails():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch", "--steps", "2", "--warmup", "0"],
                       env=_env(SS_DRY_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
    assert "rank 1 exited with code 7" in r.stderr


def test_external_launcher_form_still_works():
    """The driver's form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", BENCH, "--gpus", "2", "--dry-launch",
                        "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["pairs_counted"] == 2 * 3


def test_parent_makes_no_gpu_call_before_spawning():
    """Static check of the launcher branch: nothing between argument parsing and launch_ranks() may touch torch.cuda."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    upto = main[:main.index("launch_ranks(args.gpus")]
    assert "torch.cuda" not in upto and "_lib.load" not in upto
    launcher = src[src.index("def launch_ranks"):src.index("def dry_step_factory")]
    launcher = launcher[launcher.index('"""', launcher.index('"""') + 3):]          # code only, not the docstring
    assert "torch.cuda" not in launcher and "os.exec" not in launcher and "execv" not in launcher


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_through_the_real_step():
    """On the GPU box: `python bench.py --gpus 2` with SS_DIST_BACKEND=gloo (RCCL refuses two ranks on one device): the
    self-launched ranks run the REAL hot-segment step on cuda:0, meet at the barriers, reduce the metrics and rank 0 prints
    one line with n_gpus = 2 and twice the pairs of a single rank.  (The rate of such a run means nothing.)"""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "1", "--no-cpu-baseline",
           "--no-other-engines", "--height", "256", "--width", "256", "--maxdisp", "64"]
    def run_once():
        # own process group: on a timeout the launcher AND its ranks are killed (ranks orphaned on the GPU would slow down
        # everything that runs after this test)
        p = subprocess.Popen(cmd, env=_env(SS_DIST_BACKEND="gloo"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             start_new_session=True)
        try:
            out, err = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.communicate()
            return None
        return subprocess.CompletedProcess(cmd, p.returncode, out, err)

    r = run_once() or run_once()    # (the run takes ~5 s; one retry for a rendezvous that does not come up on a box that is still paging in)
    assert r is not None, "two attempts timed out"
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    res = lines[0]
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["scaling"] == "weak"
    assert abs(res["value"] * res["ms_per_step"] * 1e-3 - 2.0) < 1e-4          # pairs/s x s/step = 2 pairs per step over both ranks (the line carries 6 significant digits)
    assert "roofline" in res
