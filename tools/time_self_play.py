"""End-to-end main.self_play(n) timing (host bookkeeping, sample emission and augmentation included):
    python tools/time_self_play.py [games] [sims] [blocks] [device_replay 0/1] [carry_over calls]
First the synchronous call (one self_play(games), traced by the number of active games), then -- carry_over calls > 0, default 3 --
configure(carry_over=True) and that many consecutive self_play(games) calls after two untimed ones (the steady state of a
training loop: every call returns its own episodes, later calls' games keep the slots busy). carry_over calls = 0: only the
synchronous part; a negative number: only the carry-over part with that many calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from alpha_omok_amd import main
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 400
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dr = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
carry = int(sys.argv[5]) if len(sys.argv) > 5 else 3


def carry_part(carry):
    main.configure(board_size=9, n_mcts=sims, n_blocks=nb, seed=0, device_replay=dr, carry_over=True)
    main.self_play(n)                 # untimed: fills the pipeline (the first call starts all its games at once,
    main.self_play(n)                 # the second still sees that wave of simultaneous game ends)
    main.cur_memory.clear(); main.rep_memory.clear()
    t0 = time.perf_counter()
    per = []
    for _ in range(carry):
        t1 = time.perf_counter()
        r = main.self_play(n)
        per.append((r['moves'], time.perf_counter() - t1))
    dt = time.perf_counter() - t0
    moves = len(main.cur_memory)
    print("carry-over: %d x self_play(%d) @%d sims, %d blocks, device_replay=%s: %.1f s, %d move decisions returned = %.0f move-decisions/s "
          "(per call: %s); %d games of later calls in flight" % (carry, n, sims, nb, dr, dt, moves, moves / dt,
          ", ".join("%.0f" % (m / t) for m, t in per), int(main._pool.active.sum())))


if carry < 0:
    carry_part(-carry)
    sys.exit(0)
main.configure(board_size=9, n_mcts=sims, n_blocks=nb, seed=0, device_replay=dr, carry_over=False)
main.self_play(min(n, 64))            # warm-up: builds the engine, exports the net
main.cur_memory.clear(); main.rep_memory.clear()
os.environ["AO_SELFPLAY_TRACE"] = "1"
t0 = time.perf_counter()
main.self_play(n)
dt = time.perf_counter() - t0
moves = len(main.cur_memory)
print("self_play(%d) @%d sims, %d blocks, device_replay=%s: %.1f s, %d move decisions = %.0f move-decisions/s, "
      "%d replay entries" % (n, sims, nb, dr, dt, moves, moves / dt, len(main.rep_memory)))

tr = main.last_trace
if tr:
    ts = sum(t for _, t in tr)
    print("searches: %d, %.1f s inside search+play (%.1f s outside: sample assembly, augmentation, bookkeeping)" % (len(tr), ts, dt - ts))
    for lo, hi in ((3072, 1 << 30), (1024, 3072), (256, 1024), (48, 256), (0, 48)):
        sel = [(a, t) for a, t in tr if lo <= a < hi]
        if sel:
            print("  active games in [%d, %s): %3d searches, %6.2f s, %7d move decisions, %.0f move-decisions/s" % (
                lo, "inf" if hi > 1 << 20 else hi, len(sel), sum(t for _, t in sel), sum(a for a, _ in sel),
                sum(a for a, _ in sel) / sum(t for _, t in sel)))

if carry > 0:
    os.environ.pop("AO_SELFPLAY_TRACE", None)
    main.release_engine()
    carry_part(carry)
