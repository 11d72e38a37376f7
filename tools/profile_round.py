#!/usr/bin/env python3
"""Collects the rocprofv3 evidence bench.py's roofline refers to. Run ON THE GPU BOX:

    python tools/profile_round.py r1e            # writes gpurun_out/profiles_r1e/*

  <tag>_kernel_stats.txt   rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1`
  <tag>_pmc.txt            FETCH_SIZE / WRITE_SIZE / SQ busy counters, one pass per counter set
                           (separate runs with --kernel-trace only, as the pool requires)
  <tag>_traffic.json       HBM bytes per launch of the dominant kernel (FETCH_SIZE x2 on gfx950)
  <tag>_single_game_kernel_stats.txt   the latency path (one game, 400 sims/move)
Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch)."""
import glob, json, os, sqlite3, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1x"
out = os.path.join(REPO, "gpurun_out", "profiles_" + tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
summ = os.path.join(REPO, "tools", "rocpd_summary.py")


def prof(name, extra, cmd):
    d = os.path.join(out, "raw_" + name)
    subprocess.run(["rocprofv3", "--kernel-trace"] + extra + ["-d", d, "-o", name, "--"] + cmd, cwd="/tmp", env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return dbs[0] if dbs else None


bench = ["python", os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-single-game", "--no-fp32-compare", "--no-ten-block", "--no-tictactoe", "--no-trained-net", "--no-wide-board"]
db = prof("stats", ["--stats"], bench + ["--steps", "2", "--warmup", "1"])
with open(os.path.join(out, tag + "_kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block\n")
    f.write(subprocess.run([sys.executable, summ, "stats", db], capture_output=True, text=True).stdout)

db1 = prof("single", ["--stats"], ["python", os.path.join(REPO, "tools", "time_single_game.py")])
with open(os.path.join(out, tag + "_single_game_kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/time_single_game.py   (1 game, 9x9, 400 sims/move, 4-block net)\n")
    f.write(subprocess.run([sys.executable, summ, "stats", db1], capture_output=True, text=True).stdout)

short = bench + ["--steps", "1", "--warmup", "0", "--sims", "20"]
sets = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
        "sq": ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES"]}
vals = {}
with open(os.path.join(out, tag + "_pmc.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 0 --sims 20 "
            "--no-cpu-baseline --no-single-game --no-fp32-compare --no-ten-block (one pass per counter set)\n"
            "# FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE under-counts wide coalesced reads "
            "by 2x (MI355X_MICROARCH.md, HBM section)\n")
    for name, ctrs in sets.items():
        dbp = prof("pmc_" + name, ["--pmc"] + ctrs, short)
        f.write("## pmc_%s\n" % name)
        if not dbp:
            f.write("(no output)\n")
            continue
        f.write(subprocess.run([sys.executable, summ, "pmc", dbp], capture_output=True, text=True).stdout)
        c = sqlite3.connect(dbp)
        for k, cn, a in c.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                  "group by kernel_name, counter_name"):
            vals[(k, cn)] = a

c = sqlite3.connect(db)
row = c.execute("select name, avg(end-start), count(*) from kernels group by name order by sum(end-start) desc limit 1").fetchone()
dom_name = row[0] if row else None
dom = [k for (k, cn) in vals if dom_name and k == dom_name]
if dom:
    k = dom[0]
    short = k.replace("void ", "").replace("ao::", "").split("(")[0]
    fetch, write = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
    busy, mfma = vals.get((k, "SQ_BUSY_CYCLES")), vals.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"))
    tj = {"kernel": short,
          "workload": "4096 boards, 4-block/128-ch PVNet, one launch = the trunk convs + heads of the dominant kernel",
          "fetch_size_kib": fetch, "write_size_kib": write, "fetch_correction": 2.0,
          "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if fetch is not None and write is not None else None,
          "source": "profiles/%s_pmc.txt (rocprofv3 --pmc, separate passes, MI355X)" % tag}
    # MFMA pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs; the kernel offers
    # 1024 x (duration x 2.4 GHz) SIMD-cycles (duration from the --stats pass of the same build)
    if row and row[1] and mfma:
        tj["avg_launch_ns"] = row[1]
        tj["mfma_busy_fraction"] = mfma / (1024.0 * row[1] * 2.4)
    # the same against the cycles the kernel really ran (GRBM_GUI_ACTIVE is summed over the 8 XCDs): the clock
    # drops under the power limit, so this is the pipe utilisation proper
    gui = vals.get((k, "GRBM_GUI_ACTIVE"))
    if gui and mfma:
        tj["mfma_busy_fraction_of_cycles"] = mfma / (1024.0 * gui / 8.0)
        if row and row[1]:
            tj["effective_clock_ghz"] = (gui / 8.0) / row[1]
    sys.path.insert(0, REPO)
    from alpha_omok_amd.build import source_hash
    tj["csrc_sha16"] = source_hash()
    tree = [k for (k, cn) in vals if "k_expand_select" in k and cn == "FETCH_SIZE"]
    if tree:
        tf, tw = vals.get((tree[0], "FETCH_SIZE")), vals.get((tree[0], "WRITE_SIZE"))
        trow = c.execute("select avg(end-start), count(*) from kernels where name like '%k_expand_select%'").fetchone()
        if tf is not None and tw is not None:
            # narrow (4-8 B per lane) scattered accesses: the 2x FETCH_SIZE correction of the guide is calibrated for wide
            # streaming reads only, so both readings are kept; `hbm_bytes_per_launch` uses the corrected one (upper bound)
            tj["tree"] = {"kernel": "k_expand_select", "fetch_size_kib": tf, "write_size_kib": tw,
                          "hbm_bytes_per_launch": (2.0 * tf + tw) * 1024.0,
                          "hbm_bytes_per_launch_uncorrected_fetch": (tf + tw) * 1024.0,
                          "avg_launch_ns": trow[0] if trow else None}
    tj["sq_valu_mfma_busy_cycles"] = mfma
    tj["sq_busy_cycles"] = busy
    tj["grbm_gui_active"] = vals.get((k, "GRBM_GUI_ACTIVE"))
    with open(os.path.join(out, tag + "_traffic.json"), "w") as f:
        json.dump(tj, f, indent=1)
print("written:", sorted(os.listdir(out)))
