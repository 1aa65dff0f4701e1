"""GPU (one MI355X): the multi-rank path of bench.py rehearsed with two torch-free ranks that share the one GPU (the 4 doubles per bead over
the loopback socket hub; the driver's 8-GPU run uses ncclAllGather inside the library on one rank per GPU).  The whole-job result must carry the contract's fields and the
ensemble potential must equal the single-process value bit for bit (the combine adds the per-bead terms in bead order)."""
import json
import os
import subprocess
import sys

import pytest

import util
from mpmcxx_amd import ranks

pytestmark = pytest.mark.gpu
ARGS = ["--beads", "4", "--natoms", "1000", "--steps", "2", "--warmup", "1", "--cpu-baseline", "none"]


def last_json(txt):
    return json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])


def run(cmd, env):
    return subprocess.run(cmd, cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MPMC_RDZV_FILE")}


def one_rocm(rank_info):
    """a rank process that carries ONE ROCm: one libamdhip64, one librccl, no torch (config.ranks[*].rocm_libs is /proc/self/maps)."""
    libs = rank_info["rocm_libs"]
    return (not rank_info["torch_imported"] and "libtorch_hip" not in libs and len(libs["libamdhip64"]) == 1 and len(libs.get("librccl", [])) == 1
            and len(libs.get("librocm_smi64", [])) <= 1)


@pytest.fixture(scope="module")
def single():
    one = run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, clean_env())
    assert one.returncode == 0, one.stderr[-2000:]
    return last_json(one.stdout)


def test_bare_multi_gpu_command_starts_its_own_torch_free_ranks(single):
    """`python3 bench.py --gpus 2 ...` with no launcher: the parent starts the two ranks itself (plain children, never an exec), relays rank
    0's line and the exit code; the ranks carry one ROCm each; same ensemble potential as one process, bit for bit."""
    two = run([sys.executable, "bench.py", "--gpus", "2", "--combine-impl", "hub", "--force-device", "0"] + ARGS, clean_env())
    assert two.returncode == 0, two.stderr[-2000:]
    a, b = single, last_json(two.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in a and k in b, k
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["value"] > 0 and b["config"]["beads_per_gpu"] == 2
    assert b["config"]["world_size"] == 2 and "bench.py itself" in b["config"]["launch"] and "hub" in b["config"]["job_channel"]
    rk = b["config"]["ranks"]
    assert len(rk) == 2 and {r["rank"] for r in rk} == {0, 1} and len({r["pid"] for r in rk}) == 2
    assert all(one_rocm(r) for r in rk), rk
    assert one_rocm(a["config"]["ranks"][0])
    assert b["instrumented_in_timed_region"] is False and a["instrumented_in_timed_region"] is False
    assert a["V_mean_K"] == b["V_mean_K"] and a["obs_rd_es_pol_vdw"] == b["obs_rd_es_pol_vdw"]  # same beads, same per-bead energies, summed in bead order


def test_ranks_under_an_external_launcher_stay_torch_free(single):
    """the driver's form: python -m torch.distributed.run ... bench.py --gpus 2.  The launcher is a torch program; the ranks are not."""
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(ranks.free_port()),
               "bench.py", "--gpus", "2", "--combine-impl", "hub", "--force-device", "0"] + ARGS, dict(clean_env(), HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert two.returncode == 0, two.stderr[-2000:]
    b = last_json(two.stdout)
    assert b["n_gpus"] == 2 and "external launcher" in b["config"]["launch"] and all(one_rocm(r) for r in b["config"]["ranks"])
    assert single["V_mean_K"] == b["V_mean_K"] and single["obs_rd_es_pol_vdw"] == b["obs_rd_es_pol_vdw"]


def test_the_cabi_communicator_on_one_shared_device_falls_back_together(single):
    """default --combine-impl cabi with both ranks forced onto the one GPU: RCCL refuses two ranks on one device (an ERROR, not a hang), every
    rank votes, the job runs with the hub's host all-gather and says so; on two real devices the communicator comes up instead."""
    two = run([sys.executable, "bench.py", "--gpus", "2", "--force-device", "0", "--comm-init-timeout", "60"] + ARGS, clean_env())
    assert two.returncode == 0, two.stderr[-2000:]
    b = last_json(two.stdout)
    assert ("cabi_comm_failed" in b and "FALL-BACK" in b["config"]["combine_impl"]) or "mpmc_pi_gather_beads" in b["config"]["combine_impl"]
    assert single["V_mean_K"] == b["V_mean_K"]


def test_the_opt_in_torch_path_still_works(single):
    two = run([sys.executable, "bench.py", "--gpus", "2", "--combine-impl", "torch", "--dist-backend", "gloo", "--force-device", "0"] + ARGS, clean_env())
    assert two.returncode == 0, two.stderr[-2000:]
    b = last_json(two.stdout)
    assert b["config"]["job_channel"] == "torch.distributed" and all(r["torch_imported"] for r in b["config"]["ranks"])
    assert all(len(r["rocm_libs"]["libamdhip64"]) == 1 for r in b["config"]["ranks"])  # torch first: the library binds to PyTorch's runtime, still one
    assert single["V_mean_K"] == b["V_mean_K"]


def test_two_ranks_on_two_devices_over_rccl(single):
    from mpmcxx_amd import energy

    if energy.device_count() < 2:
        pytest.skip("needs two GPUs (ncclCommInitRank with two ranks)")
    two = run([sys.executable, "bench.py", "--gpus", "2"] + ARGS, clean_env())
    assert two.returncode == 0, two.stderr[-2000:]
    b = last_json(two.stdout)
    assert "mpmc_pi_gather_beads" in b["config"]["combine_impl"] and "cabi_comm_failed" not in b
    assert all(one_rocm(r) and r["comm_n_ranks"] == 2 for r in b["config"]["ranks"])
    assert single["V_mean_K"] == b["V_mean_K"]


def test_inprocess_launch_drives_the_devices_through_mpmc_pi_allreduce():
    """--launch inprocess: ONE process, bead b on device b mod N, mpmc_pi_allreduce (host thread per device, ncclCommInitAll).  On a one-GPU
    box both "devices" are virtual devices of device 0 (--force-device: the library's test hook): two host threads, the real hand-off and
    ordered sum, a host copy where RCCL (one rank per physical device) would be."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    inp = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--launch", "inprocess", "--force-device", "0"] + ARGS,
                         cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert inp.returncode == 0, inp.stderr[-2000:]
    a, b = last_json(one.stdout), last_json(inp.stdout)
    assert b["n_gpus"] == 2 and b["config"]["world_size"] == 1 and b["config"]["launch"].startswith("inprocess")
    assert "mpmc_pi_allreduce" in b["config"]["combine_impl"]
    assert len(b["config"]["ranks"]) == 2 and all(r["comm_n_ranks"] == 2 and r["distinct_devices"] == 2 for r in b["config"]["ranks"])
    assert a["V_mean_K"] == b["V_mean_K"] and a["obs_rd_es_pol_vdw"] == b["obs_rd_es_pol_vdw"]


def test_inprocess_launch_on_two_devices():
    from mpmcxx_amd import energy

    if energy.device_count() < 2:
        pytest.skip("needs two GPUs (one process driving two devices: ncclCommInitAll with two ranks)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + ARGS, cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    inp = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--launch", "inprocess"] + ARGS,
                         cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0 and inp.returncode == 0, (one.stderr[-1000:], inp.stderr[-2000:])
    a, b = last_json(one.stdout), last_json(inp.stdout)
    assert all(r["comm_n_ranks"] == 2 and r["distinct_devices"] == 2 for r in b["config"]["ranks"])
    assert a["V_mean_K"] == b["V_mean_K"]
