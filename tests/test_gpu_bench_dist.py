"""GPU: bench.py's N > 1 branch, run as the driver runs it -- `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` -- on the one GPU of the test box: both ranks on device 0 (JK_BENCH_ONE_DEVICE=1) and gloo for the few integers
they exchange (RCCL wants a device per rank), at sizes that take seconds.  Checks the JSON contract of the three workload
lines (n_gpus, totals, weak scaling) and that the lane shards of the ranks add up to the single-process job.  A second
test runs world size 1 over the "nccl" backend (JK_BENCH_FORCE_DIST=1): RCCL's init and the count / seed-offset all-gathers
on device tensors execute on hardware once.  The children are fresh processes (started before they touch the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = {
    "illumina": ["--pairs", "150000", "--lanes", "8192", "--genome-mbp", "3"],
    "hap": ["--lanes", "8192", "--genome-mbp", "0.2"],                     # 24 x 250 kb, 8 haplotypes, 75 k pairs per rank
    "pacbio": ["--lanes", "4096", "--genome-mbp", "6"],                     # 12 k reads per rank
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(workload, n, extra_env, tmp_path):
    env = dict(os.environ)
    env.update(extra_env)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--workload", workload] + SMALL[workload]
    if n > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:
        env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", str(_free_port())
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    (tmp_path / ("%s_n%d.out" % (workload, n))).write_text(r.stdout + "\n---- stderr ----\n" + r.stderr[-4000:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["illumina", "hap", "pacbio"])
def test_bench_two_ranks_under_torch_distributed_run(workload, tmp_path):
    one = run_bench(workload, 1, {}, tmp_path)
    two = run_bench(workload, 2, {"JK_BENCH_ONE_DEVICE": "1", "JK_BENCH_BACKEND": "gloo"}, tmp_path)
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
        assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
        assert d["value"] > 0 and d["ms_per_step"] > 0 and set(d["step_ms"]) == {"min", "median", "max"}
        assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"
        assert "workload" in d["config"] and "model" not in d["config"]
    assert two["metric"] == one["metric"] and two["unit"] == one["unit"]
    # weak scaling: every rank brings a full per-GPU job; value = units of ALL ranks / time, so two ranks sharing one device
    # cannot be slower than half nor faster than twice the single-process rate
    assert 0.4 * one["value"] < two["value"] < 2.5 * one["value"]


def test_world_size_one_over_rccl(tmp_path):
    """backend "nccl" (= RCCL): process-group init, the seed-offset all-gather of open_shard and the count all-gather of
    exchange_counts on device tensors, the MAX all-reduce of the elapsed time, the barriers."""
    d = run_bench("illumina", 1, {"JK_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"}, tmp_path)
    assert d["n_gpus"] == 1 and d["value"] > 0
