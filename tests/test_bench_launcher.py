"""bench.py as the driver invokes it (`python bench.py --gpus N ...`, no external launcher): the script must spawn one
rank per GPU itself, refuse to report a smaller job under a bigger name, and gather one record per rank.
CPU part: DCREG_BENCH_DRYRUN=1 replaces the device work by a synthetic record so that launcher, rendezvous (gloo), fence,
max-over-ranks and gather run here.  GPU part (-m gpu): the real workload with two ranks sharing the box's one device over
gloo (control flow only; RCCL with one rank per GPU is what the 8-GPU node runs), both sharding modes."""
import json
import os
import subprocess
import sys

import pytest

import helpers as h

BENCH = os.path.join(h.REPO, "bench.py")


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, cwd=h.REPO, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])       # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_self_spawn_two_ranks_dry_run():
    p = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5"], {"DCREG_BENCH_DRYRUN": "1", "DCREG_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p)
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["steps"] == 20 and j["warmup"] == 5
    assert j["ranks"] == [0, 1] and j["rank_seeds"] == [100, 101]     # one record per rank, every rank its own pair
    assert all(t >= 0.02 for t in j["block_times_s"])                 # max over ranks: rank 1 sleeps twice as long as rank 0
    _check_c5_leg(j, 2)


def _check_c5_leg(j, world):
    """the strong-scaling leg of a multi-rank run: every trial of the 5000 ran on exactly one rank (k = rank mod world), the records of
    all ranks reached rank 0 through the one gather, the statistics are those of the whole experiment"""
    c5 = j["c5_montecarlo_5000"]
    assert c5["n_gpus"] == world and c5["scaling"] == "strong" and c5["trials"] == 5000 and c5["rccl_ranks_seen"] == world
    assert c5["iterations"] == sum(10 + k % 7 for k in range(5000)) and c5["converged_runs"] == sum(1 for k in range(5000) if k % 4)


@pytest.mark.timeout(600)
def test_self_spawn_eight_ranks_dry_run():
    """The shape the driver's scaling run has - `python bench.py --gpus 8` on one node: eight ranks rendezvous on 127.0.0.1, fence,
    take the max over ranks and gather one record each; every rank gets its own scan-pair seed and its share of the host's CPUs."""
    p = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5"], {"DCREG_BENCH_DRYRUN": "1", "DCREG_BENCH_BACKEND": "gloo"}, timeout=550)
    assert p.returncode == 0, p.stderr[-2000:]
    j = _json_line(p)
    assert j["dry_run"] is True and j["n_gpus"] == 8
    assert j["ranks"] == list(range(8)) and j["rank_seeds"] == [100 + r for r in range(8)]
    assert all(t >= 0.08 for t in j["block_times_s"])                 # max over ranks: rank 7 sleeps 8 x 10 ms
    assert j["host_threads_per_rank"] >= 1 and len(set(j["host_threads"])) == 1 and j["host_threads"][0] == j["host_threads_per_rank"]
    # the share is taken from the CPUs the job was started with, not from a rank's slice of them
    from dcreg_amd import hostinfo
    assert j["host_threads_per_rank"] == hostinfo.threads_per_rank(8)
    _check_c5_leg(j, 8)


def test_refuses_a_job_it_cannot_run():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    p = _bench(["--gpus", "2"], {})
    assert p.returncode != 0 and "--gpus 2 requested" in p.stderr and "{" not in p.stdout
    p = _bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("sharding", ["pairs", "points"])
def test_two_ranks_on_one_device_real_workload(sharding):
    p = _bench(["--gpus", "2", "--steps", "20", "--warmup", "20", "--repeats", "3", "--min-seconds", "0", "--workload", "c2_cylinder_100k", "--sharding", sharding,
                "--no-cpu-baseline", "--no-configs", "--concurrent-pairs", "0"],
               {"DCREG_BENCH_BACKEND": "gloo", "DCREG_BENCH_LOCAL_RANK": "0"}, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p)
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["repeats"] == 3 and j["value"] > 0
    assert j["scaling"] == ("weak" if sharding == "pairs" else "strong")
    assert j["config"]["rank_pair_seeds"] == [100, 101]
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["final_stats"]["mean_trans_error_m"] < 0.01
    if sharding == "points":                                          # every rank holds a slice: half the queries per launch
        assert j["roofline"]["algorithmic_bytes_per_launch"] == 72 * 50_000


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_run_the_montecarlo_experiment_as_one_job():
    """N > 1 without extra flags: after the weak-scaling measurement of the main workload the 5000-trial Monte-Carlo experiment runs ONCE
    across the ranks (trials k = rank mod N), its trial records are gathered inside the timed region and rank 0 takes the statistics -
    here with two ranks sharing the box's one device over gloo."""
    p = _bench(["--gpus", "2", "--steps", "30", "--warmup", "30", "--repeats", "2", "--min-seconds", "0", "--workload", "c1_fixture_7562", "--no-cpu-baseline",
                "--concurrent-pairs", "0"], {"DCREG_BENCH_BACKEND": "gloo", "DCREG_BENCH_LOCAL_RANK": "0"}, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    j = _json_line(p)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and list(j["configs"]) == ["c5_montecarlo_5000"]
    c5 = j["configs"]["c5_montecarlo_5000"]
    assert c5["n_gpus"] == 2 and c5["scaling"] == "strong" and c5["rccl_ranks_seen"] == 2
    st = c5["montecarlo"]["statistics"]
    assert st["total_runs"] == 5000 and 0.5 < st["success_rate"] < 0.95
    # all trials, not one rank's share: the iterations of one experiment
    assert 60_000 < c5["icp_iterations_per_step"] < 100_000 and c5["value"] > 0


def test_a_failing_montecarlo_job_becomes_an_error_record():
    """bench.py at N > 1 runs the Monte-Carlo leg as a job of its own (montecarlo_child): a job that fails - here because the box has fewer
    than two devices, so the child refuses to start - comes back as an error record for the JSON line, not as an exception that would take
    the headline measured before it along."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    sys.path.insert(0, h.REPO)
    import bench
    args = bench.parse_args(["--gpus", "2"])
    env_keep = {k: os.environ.pop(k) for k in ("DCREG_BENCH_BACKEND", "DCREG_BENCH_DRYRUN") if k in os.environ}
    try:
        rec = bench.montecarlo_child(args, 2, timeout_s=300.0)
    finally:
        os.environ.update(env_keep)
    assert "error" in rec and "rc" in rec["error"] and "--gpus 2 requested" in rec["stderr_tail"]
