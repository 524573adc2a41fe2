#!/usr/bin/env python3
"""Which branch of a block is the critical one?  From a rocprofv3 --kernel-trace csv of bench.py: for every SS_Conv_SSM block of the
LAST training step, forward and backward, the wall time between the block's split and its shuffle (MedMamba.py:350-357) and the
kernel time each HSA queue (main stream = SS2D branch, side stream = conv branch) spent inside that window, plus the idle time of
each queue inside it.  A block whose side queue is busy to the end of the window while the main queue idles waits for the conv
branch, and the other way round.  usage: tools/trace_block_balance.py DIR [nscan]"""
import csv, glob, os, re, sys


def main():
    d = sys.argv[1]; nscan = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if re.search(r"scan_fwd(_wg|_rows)?_kernel", r["Kernel_Name"])]
    last = rows[idx[-nscan]:]
    # the step starts a little before the first forward scan: take everything from the last patch-embed LayerNorm kernel on
    first = idx[-nscan]
    for i in range(first, max(0, first - 60), -1):
        if "nchw_ln_rows_fwd" in rows[i]["Kernel_Name"]:
            last = rows[i:]
            break
    t0 = int(last[0]["Start_Timestamp"])
    S = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e3
    E = lambda r: (int(r["End_Timestamp"]) - t0) / 1e3
    queues = sorted({r.get("Queue_Id", "?") for r in last})
    main_q = max(queues, key=lambda q: sum(1 for r in last if r.get("Queue_Id", "?") == q and "scan_" in r["Kernel_Name"]))

    def busy(q, a, b):
        iv = sorted((max(a, S(r)), min(b, E(r))) for r in last if r.get("Queue_Id", "?") == q and E(r) > a and S(r) < b)
        tot, end = 0.0, a
        for s, e in iv:
            if e > end:
                tot += e - max(s, end); end = e
        last_end = max((e for _, e in iv), default=a)
        return tot, last_end

    def windows(open_pat, close_pat):
        opens = [r for r in last if re.search(open_pat, r["Kernel_Name"])]
        closes = [r for r in last if re.search(close_pat, r["Kernel_Name"])]
        out, j = [], 0
        for o in opens:
            while j < len(closes) and S(closes[j]) < S(o):
                j += 1
            if j < len(closes):
                out.append((S(o), E(closes[j]))); j += 1
        return out

    print(f"# source {f}; main queue q{main_q}; times in us inside the last step")
    for name, wins in (("forward  (block_split_fwd .. shuffle_residual_fwd)", windows(r"block_split_fwd_kernel", r"shuffle_residual_fwd_kernel")),
                       ("backward (shuffle_residual_bwd .. block_split_bwd)", windows(r"shuffle_residual_bwd_kernel", r"block_split_bwd_kernel"))):
        print(f"\n## {name}")
        print(f"{'block':>5} {'start':>9} {'wall':>8} | {'main busy':>9} {'main ends':>9} | {'side busy':>9} {'side ends':>9} | waits for")
        tw = tm = ts = 0.0
        for n, (a, b) in enumerate(wins):
            line = f"{n:>5} {a:>9.0f} {b - a:>8.0f} |"
            ends = {}
            for q in queues:
                bz, le = busy(q, a, b)
                ends[q] = le
                line += f" {bz:>9.0f} {le - a:>9.0f} |"
                if q == main_q: tm += bz
                else: ts += bz
            side_q = [q for q in queues if q != main_q]
            # the closing kernel runs on the main queue: which queue finished its branch work later (before the closing kernel)?
            closing_start = max((S(r) for r in last if r.get("Queue_Id", "?") == main_q and S(r) < b and E(r) >= b - 1e-6), default=b)
            mb, _ = busy(main_q, a, closing_start)
            m_end = max((E(r) for r in last if r.get("Queue_Id", "?") == main_q and E(r) <= closing_start + 1e-6 and E(r) > a), default=a)
            s_end = max((E(r) for q in side_q for r in last if r.get("Queue_Id", "?") == q and E(r) <= closing_start + 1e-6 and E(r) > a), default=a)
            line += f" {'conv branch' if s_end > m_end + 2 else 'SS2D branch'} (main {m_end - a:.0f}, side {s_end - a:.0f})"
            tw += b - a
            print(line)
        print(f"  sum: wall {tw:.0f} us, main busy {tm:.0f} us, side busy {ts:.0f} us")


if __name__ == "__main__":
    main()
