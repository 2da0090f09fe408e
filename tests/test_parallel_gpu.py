"""The product's data-parallel iteration on real kernels, as a multi-process GPU test: `tests/dp_rehearsal.py`
started under `torch.distributed.run` as CHILD processes (fresh interpreters; nothing is exec'ed in this process).

  * two gloo ranks sharing the one card of a single-GPU box: cases A1 / A2 / B / C of the rehearsal (gradient mean all-reduce,
    all-gathered contrastive negatives, graph segments with the collectives as eager seams) against single-process references;
  * the same over RCCL (`nccl`, one rank per GPU) wherever the box has at least two GPUs -- so that a multi-GPU driver box runs the
    RCCL path under a correctness check BEFORE it benches it.  Skipped on a one-GPU box.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rehearse(backend, nproc, tmp_path):
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = tmp_path / f"dp_{backend}.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XMC_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", XMC_DUMP_AFTER="420")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_rehearsal.py"), "--out", str(out)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=480)
    assert r.returncode == 0, r.stdout[-3000:]
    rep = json.loads(out.read_text())
    assert rep["world"] == nproc and rep["backend"] == backend
    assert set(rep["cases"]) == {"A1_bench_cfg_D_phase", "A2_no_batch_coupled_terms", "B_gather_negatives", "C_graph_segments_equal_eager"}
    worst = max(v for c in rep["cases"].values() for v in c.values())
    assert worst <= 2e-3, rep["cases"]          # (the script asserts the same per case)
    print(f"\n[dp rehearsal, {nproc} x {backend}] worst relative difference {worst:.2e}: {json.dumps(rep['cases'])}")
    return rep


def test_two_gloo_ranks_on_one_card_equal_single_process_references(tmp_path):
    _rehearse("gloo", 2, tmp_path)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has fewer than two")
def test_two_rccl_ranks_equal_single_process_references(tmp_path):
    _rehearse("nccl", 2, tmp_path)


def _bench(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert r.returncode == 0 and line, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(line[-1])


def test_bench_launches_two_ranks_itself_and_reports_the_job():
    """`python bench.py --gpus 2` without a launcher (the path the driver's N > 1 runs take when it calls the script directly): the
    script starts its ranks as a child `torch.distributed.run`, one process per GPU; here two gloo ranks share the one card
    (XMC_DIST_BACKEND=gloo).  The line must describe the JOB: n_gpus 2, global batch 2 x per-GPU batch, the collectives of an
    iteration as graph seams (D's gradients in two parts -- the head's and last blocks' all-reduce started before the rest of the backward,
    finished after it -- and G's gradients: 3; with all-gathered negatives one gather in the D step (the sentence embeddings the labels
    are made from, the image and the text embeddings, in one collective) and one in the G step more: 5), their host time, and an aggregate rate below the one-rank rate (two ranks time-share one card and every gloo all-reduce goes
    through host memory: measured x0.34 with 15.8 ms of host time per collective against a 4.6 ms iteration -- a statement about gloo
    on one card, not about RCCL)."""
    common = ["--steps", "6", "--warmup", "2", "--workload", "config2", "--batch", "32", "--no_parity", "--no_alt_precision", "--no_entrypoint",
              "--no_cpu_baseline", "--no_roofline"]
    one = _bench(["--gpus", "1"] + common, {})
    two = _bench(["--gpus", "2"] + common, {"XMC_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["global_batch"] == 64 and two["config"]["parallelism"] == "dp2"
    d = two["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["process_group"]
    assert d["collectives_per_iteration"] == {"g_step": 3}, d
    assert d["seam_host_ms"] is not None and d["seam_host_ms"] > 0
    assert one["dist"]["collectives_per_iteration"] in (None, {"g_step": 0})
    ratio = two["value"] / one["value"]
    print(f"\n[bench --gpus 2, gloo, one card] {two['value']:.0f} images/s against {one['value']:.0f} on one rank (x{ratio:.2f}); "
          f"{d['collectives_per_iteration']} collectives per iteration, {d['seam_host_ms']} ms of host time each")
    assert 0.1 <= ratio <= 1.3, ratio
    gat = _bench(["--gpus", "2", "--gather_negatives"] + common, {"XMC_DIST_BACKEND": "gloo"})
    assert gat["config"]["parallelism"] == "dp2+gather" and gat["dist"]["collectives_per_iteration"] == {"g_step": 5}, gat["dist"]


def test_rccl_at_world_size_one_runs_the_two_part_backward_under_graph_replay():
    """`bench.py --force_dp`: a one-rank `nccl` process group with the data-parallel path forced on -- the discriminator's backward in two
    calls, its first all-reduce started asynchronously on RCCL's stream between two graph segments and waited for after the second call, the
    generator's bucket, all as graph seams.  The only RCCL execution a one-GPU box allows: that the calls work under capture / replay, that
    the losses stay finite, and that it costs a one-rank job next to nothing."""
    common = ["--steps", "6", "--warmup", "2", "--workload", "config2", "--batch", "32", "--no_parity", "--no_alt_precision", "--no_entrypoint",
              "--no_cpu_baseline", "--no_roofline"]
    plain = _bench(common, {})
    forced = _bench(["--force_dp"] + common, {})
    d = forced["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["forced_at_world_1"] and d["collectives_per_iteration"] == {"g_step": 3}, d
    assert forced["config"]["losses_finite"] and forced["config"]["hipgraph"]
    ratio = forced["value"] / plain["value"]
    print(f"\n[bench --force_dp, RCCL, world 1] {forced['value']:.0f} images/s against {plain['value']:.0f} without collectives (x{ratio:.2f}); "
          f"{d['seam_host_ms']} ms of host time per collective call")
    assert ratio >= 0.85, ratio


def test_two_ranks_through_the_entry_point_keep_identical_weights(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 ... xmc_gan/train_gan.py`-style: `main()` on two gloo ranks sharing the card
    (tests/entry_dp_rehearsal.py): rank 0's initial weights broadcast, each rank its own synthetic batches (seed + rank), `train()` replaying graph
    segments with the three gradient collectives of an iteration as seams.  After five iterations the two ranks hold the SAME weights bit for
    bit (the all-reduced gradients are the same numbers on both, Adam is element-wise) -- and not the weights a single process reaches from
    that seed (the other rank's batches took part), losses finite, the graph replay in use."""
    import math
    from test_entrypoint_gpu import _mini_yml
    yml = _mini_yml(tmp_path)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(XMC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "entry_dp_rehearsal.py"), "--cfg", yml, "--out", str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=480)
    assert r.returncode == 0, r.stdout[-3000:]
    a, b = (torch.load(tmp_path / f"rank{k}.pt") for k in (0, 1))
    assert a["hipgraph"] and b["hipgraph"]
    for rep in (a, b):
        assert {"errD", "errG"} <= set(rep["losses"]) and all(math.isfinite(v) for v in rep["losses"].values()), rep["losses"]
    assert a["weights"].keys() == b["weights"].keys()
    for k, va in a["weights"].items():
        assert torch.equal(va, b["weights"][k]), k
    assert a["losses"]["errD"] != b["losses"]["errD"]          # each rank saw its own batch
    # a single process from the same seed ends elsewhere
    one = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "entry_dp_rehearsal.py"), "--cfg", yml, "--out", str(tmp_path / "one")],
                         env={**env, "RANK": "0"}, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=480)
    assert one.returncode == 0, one.stdout[-3000:]
    c = torch.load(tmp_path / "one" / "rank0.pt")
    assert any(not torch.equal(va, c["weights"][k]) for k, va in a["weights"].items())
