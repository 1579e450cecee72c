"""`python bench.py --gpus N --preflight` -- the contact check the first multi-GPU box will run before any timing (launcher respawn under
torch.distributed.run, device binding, rendezvous-id exchange, one all-reduce, two data-parallel training steps with the per-bucket ready / done
times) -- kept alive on the one-GPU box: two ranks over gloo sharing cuda:0 (RCCL refuses two ranks on one device), and the one-rank run through
the native RCCL engine.  What these runs cannot show is a collective between two DEVICES; everything before and around it they execute."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):  # bench.py starts the ranks itself
        e.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, (res.returncode, res.stdout[-2000:], res.stderr[-4000:])
    return json.loads(lines[0])


def test_two_rank_preflight_over_gloo_on_one_gpu(gpu):
    rep = _run(["--gpus", "2", "--preflight"], VMM_DIST_BACKEND="gloo")
    assert rep["preflight"] and rep["n_gpus"] == 2 and rep["ok"], rep
    ck = rep["checks"]
    assert ck["launcher"]["WORLD_SIZE"] == 2 and ck["launcher"]["backend"] == "gloo" and ck["launcher"]["MASTER_ADDR"] == "127.0.0.1"
    assert ck["all_gather"]["ok"] and [r for r, _ in ck["all_gather"]["rank_device_table"]] == [0, 1]
    assert ck["rccl_unique_id"]["same_on_every_rank"] in (True, None)  # (None: RCCL did not load on the box -- reported, not fatal)
    st = ck["train_step"]["torch"]
    assert st["ok"] and st["replica_checksum_spread"] == 0.0 and st["selfcheck"]["ok"] and st["selfcheck"]["sum_of_rank_ids"] == 1.0
    assert st["buckets"] >= 2 and len(st["bucket_spans_ms"]) == st["buckets"]
    assert all(b >= a for a, b in st["bucket_spans_ms"])
    assert "native" not in ck["train_step"]  # (two ranks on one device: the RCCL communicator is not attempted)


def test_one_rank_preflight_through_the_native_rccl_engine(gpu):
    rep = _run(["--gpus", "1", "--preflight"])
    assert rep["ok"] and rep["n_gpus"] == 1, rep
    st = rep["checks"]["train_step"]
    assert st["torch"]["ok"]
    assert st["native"]["ok"] and st["native"]["selfcheck"]["ok"] and st["native"]["buckets"] >= 2, st["native"]
