"""CPU: the measurement plumbing of bench.py (round-2 review, item 1).

* the committed rocprofv3 PMC summary bench.py's `roofline.traffic` quotes must be accepted by bench.py's own
  matching rule for the kernel that runs the default workload and for the CURRENT kernel sources -- a stale or
  mis-named profile makes the driver's record say `traffic: null` (that happened in round 2);
* `python bench.py --gpus N` without a launcher starts the ranks itself and refuses -- exit code 2, nothing on
  stdout -- when fewer than N GPUs are visible, instead of printing a 1-GPU line under an N-GPU label."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


def test_newest_traffic_profile_matches_the_kernel_that_runs():
    import bench
    from alpha_omok_amd.build import source_hash
    from alpha_omok_amd.engine import plan_kernel
    path = bench.newest_traffic_profile()
    assert path is not None, "no profiles/r*_traffic.json committed"
    with open(path) as f:
        tj = json.load(f)
    # the kernel ao_search launches for BASELINE configs[2] (4096 boards, bit planes), by the library's own planning
    kname, flop = plan_kernel(4, 5, 128, 9, 4096, in_kind=2)
    assert flop == pytest.approx(2.0 * 81 * 9 * (5 * 128 + 8 * 128 * 128) * 4096)
    assert bench.kernel_key(tj["kernel"]) == bench.kernel_key(kname), (
        "%s was collected for %s, the default workload runs %s" % (os.path.basename(path), tj["kernel"], kname.split(" (")[0]))
    if os.environ.get("AO_ALLOW_STALE_PROFILE"):   # developer switch while kernels are being edited between GPU profile runs
        return
    traffic, tree, why = bench.match_traffic(tj, kname, source_hash(), True)
    assert traffic is not None and traffic > 1e9, "bench.py would report roofline.traffic = null: %s" % why
    assert tree and tree.get("hbm_bytes_per_launch"), "no tree-kernel traffic in %s" % os.path.basename(path)


def test_kernel_key_normalises_rocprof_and_library_names():
    import bench
    assert bench.kernel_key("void ao::k_trunk16hb<9, 4>(ao::TrunkHArgs)") == ("k_trunk16hb", "9")
    assert bench.kernel_key("k_trunk16hb<9, 4> (conv1 + 8 3x3 convs ...)") == ("k_trunk16hb", "9")
    assert bench.kernel_key("k_trunk16h<9, 4>") != bench.kernel_key("k_trunk16hb<9, 4>")
    assert bench.kernel_key("k_layer16h<15> (one 3x3 conv per launch ...)") == bench.kernel_key("void ao::k_layer16h<15, 4, 4, 0>(ao::LayerHArgs)")
    assert bench.kernel_key("k_boardh<15, 2> (conv1 + 20 3x3 convs ...)") == bench.kernel_key("void ao::k_boardh<15, 2>(ao::BoardHArgs)")


def test_plan_kernel_follows_batch_size_and_input_kind():
    from alpha_omok_amd.engine import plan_kernel
    assert plan_kernel(4, 5, 128, 9, 4096, in_kind=1)[0].startswith("k_trunk16h<9, 4, 0>")
    assert plan_kernel(10, 5, 128, 9, 4096, in_kind=2)[0].startswith("k_trunk16hb<9, 4, 0>")
    assert plan_kernel(4, 5, 128, 9, 1, in_kind=1)[0].startswith("k_conv_cells_h<9, 8>")
    assert plan_kernel(4, 5, 64, 9, 1, in_kind=1)[0].startswith("k_conv_cells<9>")
    assert plan_kernel(10, 5, 128, 15, 1024, in_kind=2)[0].startswith("k_boardh<15, 2>")     # wide boards, >= 128 of them: one board per workgroup, resident in LDS
    assert plan_kernel(10, 5, 128, 15, 1024, in_kind=1)[0].startswith("k_boardh<15, 1>")
    assert plan_kernel(10, 5, 128, 15, 50, in_kind=2)[0].startswith("k_layer16h<15>")
    assert plan_kernel(10, 5, 128, 15, 1024, in_kind=2, trunk_mode=6)[0].startswith("k_layer16h<15>")
    assert plan_kernel(4, 5, 128, 9, 1024, in_kind=2)[0].startswith("k_layer16hk<9, 4>")    # medium batch: cout-pair split
    assert plan_kernel(4, 5, 128, 9, 768, in_kind=2)[0].startswith("k_layer16hk<9, 4>")
    assert plan_kernel(4, 5, 128, 9, 2048, in_kind=2)[0].startswith("k_layer16h<9>")         # (the two-workgroup form is not planned by default)
    assert plan_kernel(4, 5, 128, 9, 640, in_kind=2)[0].startswith("k_row16hk<9>")            # small batch: one workgroup per group x row x cout pair
    assert plan_kernel(4, 5, 128, 9, 48, in_kind=2)[0].startswith("k_row16hk<9>")
    assert plan_kernel(4, 5, 128, 9, 32, in_kind=1)[0].startswith("k_conv_cells_h<9, 8>")
    assert plan_kernel(4, 5, 128, 9, 640, in_kind=2, trunk_mode=6)[0].startswith("k_layer16h<9>")
    assert plan_kernel(4, 5, 128, 9, 2560, in_kind=2)[0].startswith("k_layer16h<9>")
    assert plan_kernel(4, 5, 128, 9, 1024, in_kind=2, trunk_mode=6)[0].startswith("k_layer16h<9>")   # one arithmetic for every batch size
    assert plan_kernel(4, 5, 64, 9, 4096, in_kind=1)[0].startswith("k_trunk16<9>")
    assert plan_kernel(4, 5, 128, 9, 4096, in_kind=1, trunk_mode=2)[0].startswith("k_trunk16<9>")


def test_bench_gpus_n_refuses_without_n_devices():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AO_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert r.stdout.strip() == "", "a bench line was printed although the requested GPUs are not there"
    assert "--gpus 2" in r.stderr and "visible" in r.stderr


def test_build_lists_cover_every_source_and_header_under_csrc():
    """build.HEADERS / build.SOURCES are what `_stale` watches and what `source_hash()` (-> bench `csrc_sha16`, the key that
    lets a profiles/r*_traffic.json speak for a bench run) covers. Round 4 shipped net_layer_ksplit.hpp outside HEADERS: an
    edit of that header alone would not have rebuilt net.o, and the hash was blind to two kernels. Every file a translation
    unit (transitively) includes by name, and every .hip / .hpp on disk, must be in the lists."""
    import re
    from alpha_omok_amd import build as b
    on_disk_hip = sorted(f for f in os.listdir(b.CSRC) if f.endswith(".hip"))
    on_disk_hpp = sorted(f for f in os.listdir(b.CSRC) if f.endswith(".hpp"))
    assert sorted(b.SOURCES) == on_disk_hip
    local_headers = sorted(h for h in b.HEADERS if not os.path.isabs(h))
    assert local_headers == on_disk_hpp
    assert any(os.path.isabs(h) and h.endswith(os.path.join("include", "omok_hip.h")) for h in b.HEADERS)
    inc = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    for f in on_disk_hip + on_disk_hpp:
        with open(os.path.join(b.CSRC, f)) as fh:
            for name in inc.findall(fh.read()):
                path = os.path.normpath(os.path.join(b.CSRC, name))
                assert os.path.exists(path), "%s includes %s which does not exist" % (f, name)
                listed = [os.path.normpath(h if os.path.isabs(h) else os.path.join(b.CSRC, h)) for h in b.HEADERS]
                assert path in listed, "%s includes %s, which build.HEADERS does not list" % (f, name)
    # the hash follows a header's CODE (not its comments)
    h0 = b.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h0)
    target = os.path.join(b.CSRC, "net_layer_ksplit.hpp")
    with open(target) as fh:
        text = fh.read()
    real_open = open

    def fake_open(path, *a, **k):
        import io
        if os.path.abspath(path) == target:
            return io.StringIO(text + "\nstatic int ao_hash_probe_;\n")
        return real_open(path, *a, **k)
    import builtins
    builtins.open = fake_open
    try:
        h1 = b.source_hash()
    finally:
        builtins.open = real_open
    assert h1 != h0, "source_hash() does not cover net_layer_ksplit.hpp"


def test_other_workload_traffic_is_keyed_on_sources_shape_and_kernel():
    """bench.other_workload_traffic: the configs[4]-shape record of a traffic profile is used for the `wide_board` leg / a --board 15
    run only when it was collected from the same kernel sources, for that shape, on that kernel."""
    import bench
    tj = {"csrc_sha16": "abc", "other_workloads": [{"kernel": "k_boardh<15, 2>", "hbm_bytes_per_launch": 7.0e8,
                                                    "workload": {"board": 15, "games": 1024, "blocks": 10, "planes": 128}}]}
    assert bench.other_workload_traffic(tj, "abc", 15, 1024, 10, 128, "k_boardh<15, 2> (conv1 + 20 convs ...)") == 7.0e8
    assert bench.other_workload_traffic(tj, "abd", 15, 1024, 10, 128, "k_boardh<15, 2>") is None      # other sources
    assert bench.other_workload_traffic(tj, "abc", 15, 2048, 10, 128, "k_boardh<15, 2>") is None      # other shape
    assert bench.other_workload_traffic(tj, "abc", 15, 1024, 10, 128, "k_layer16h<15>") is None       # other kernel
    assert bench.other_workload_traffic(None, "abc", 15, 1024, 10, 128, "k_boardh<15, 2>") is None
    assert bench.other_workload_traffic({"csrc_sha16": "abc"}, "abc", 15, 1024, 10, 128, "k_boardh<15, 2>") is None
