"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun in front): the parent never imports torch or touches a
GPU, starts N rank processes with the torch.distributed.run environment, relays rank 0's single JSON line and returns the worst
exit code.  Runs on the CPU: --dry-launch prints what would be started, --workload launch_check is the rendezvous + the
barrier-bracketed timing protocol over gloo with world_size 2."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_dry_launch_prints_the_rank_environments():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "7", "--warmup", "2", "--dry-launch"], capture_output=True, text=True, env=_env(), timeout=60)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dry_launch"] and d["n_ranks"] == 8 and len(d["ranks"]) == 8
    ports = {r["MASTER_PORT"] for r in d["ranks"]}
    assert len(ports) == 1 and 0 < int(ports.pop()) < 65536
    for i, r in enumerate(d["ranks"]):
        assert r["RANK"] == str(i) and r["LOCAL_RANK"] == str(i) and r["WORLD_SIZE"] == "8" and r["MASTER_ADDR"] == "127.0.0.1"
        assert r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the children get the same command line minus --dry-launch
    assert d["cmd"][1] == BENCH and d["cmd"][2:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    assert "torch" not in p.stderr.lower()


def test_launcher_parent_does_not_import_torch():
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--dry-launch'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code == 0\n"
            "assert 'torch' not in sys.modules, 'the launcher imported torch'\n" % BENCH)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=_env(), timeout=60)
    assert p.returncode == 0, p.stderr


def test_two_ranks_over_gloo_one_json_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_check", "--steps", "4"], capture_output=True, text=True,
                       env=_env(RSDSFM_DIST_BACKEND="gloo"), timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["steps"] == 4
    assert d["value"] >= 4 * 0.02 * 0.9  # MAX over ranks: rank 1 sleeps 20 ms per step, rank 0 only 10 ms


def test_a_dying_rank_fails_the_launch():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_check", "--launch-grace", "1"], capture_output=True, text=True,
                       env=_env(RSDSFM_DIST_BACKEND="gloo", RSDSFM_LAUNCH_CHECK_FAIL_RANK="1"), timeout=180)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_torchrun_style_environment_is_respected():
    # a rank started by `python -m torch.distributed.run` has WORLD_SIZE == --gpus: no second level of processes
    port = "29517"
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RSDSFM_DIST_BACKEND="gloo"))
             for r in range(2)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    d = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["master"].endswith(port)
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]


def test_watchdog_prints_the_line_when_a_section_hangs():
    """the strong-scaling sub-record runs last and under a watchdog (multi-rank RCCL cannot be exercised on the 1-GPU development
    boxes): when a rank never comes back, rank 0 still prints the line it has -- the guarded key carrying the error, exactly once --
    and every rank leaves with the watchdog's NON-ZERO code: the run is reported as failed, and the launcher relays rank 0's line
    whatever the exit codes"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--workload", "launch_check", "--tiled-timeout", "3"], capture_output=True, text=True,
                       env=_env(RSDSFM_DIST_BACKEND="gloo", RSDSFM_LAUNCH_CHECK_HANG_RANK="1"), timeout=180)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "watchdog" in d["guarded"]["error"]
