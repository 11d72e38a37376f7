#!/usr/bin/env python3
"""Extended PMC passes for the dominant kernel (wave stall breakdown, LDS, L2 / fabric requests). Run ON THE GPU BOX:

    python tools/profile_deep.py r1j        # writes gpurun_out/profiles_r1j/r1j_pmc_deep.txt
    python tools/profile_deep.py r5j --filter k_boardh,k_layer16h --bench-args "--board 15 --games 1024 --blocks 10 --no-trained-net"
    python tools/profile_deep.py r5k --filter k_layer16hk,k_row16hk --cmd "python tools/time_net.py ..."   (any command instead of bench.py)

One rocprofv3 pass per counter group (--kernel-trace --pmc only, as the pool requires); per kernel: average per
dispatch of each counter, summed over its hardware instances."""
import glob, os, sqlite3, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import shlex
argv = sys.argv[1:]
def _opt(name, default):
    if name in argv:
        i = argv.index(name)
        v = argv[i + 1]
        del argv[i:i + 2]
        return v
    return default
FILTER = _opt("--filter", "k_trunk16h,k_expand_select").split(",")
BENCH_ARGS = shlex.split(_opt("--bench-args", ""))
CMD = _opt("--cmd", None)
tag = argv[0] if argv else "r1x"
out = os.path.join(REPO, "gpurun_out", "profiles_" + tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["python", os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-single-game", "--no-fp32-compare", "--no-ten-block", "--no-tictactoe", "--no-wide-board",
       "--steps", "1", "--warmup", "0", "--sims", "20"] + BENCH_ARGS
if CMD:
    cmd = shlex.split(CMD)
GROUPS = {
    "waves": ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"],
    "issue": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES"],
    "insts": ["SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_LDS_LOAD", "SQ_INSTS_VMEM_RD"],
    "lds": ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"],
    "l2": ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_PENDING_STALL_CYCLES_sum"],
    "fabric_rd": ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_GMI_32B_sum"],
    "fabric_wr": ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_BUBBLE_sum"],
}
with open(os.path.join(out, tag + "_pmc_deep.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <group> -- %s   (one pass per group; average per dispatch, summed over instances)\n" % " ".join(cmd))
    for name, ctrs in GROUPS.items():
        d = os.path.join(out, "raw_deep_" + name)
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["-d", d, "-o", name, "--"] + cmd, cwd="/tmp", env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
        f.write("## %s: %s\n" % (name, " ".join(ctrs)))
        if not dbs:
            f.write("(no output)\n")
            continue
        c = sqlite3.connect(dbs[0])
        try:
            rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                             "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        except Exception as e:
            f.write("(query failed: %r)\n" % (e,))
            continue
        for k, cn, a, n in rows:
            if any(x in k for x in FILTER):
                f.write("%-44s %-34s %6d %16.1f\n" % (k.replace("void ", "").replace("ao::", "")[:44], cn, n, a))
print(open(os.path.join(out, tag + "_pmc_deep.txt")).read())
