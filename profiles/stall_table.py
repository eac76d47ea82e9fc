"""Reduce two rocprofv3 --pmc counter_collection.csv files (SQ stall pass, TCP pass; tools/profile_stalls.sh) into a per-kernel
stall table.  MI355X_MICROARCH.md, 'rocprofv3 PMC slots': SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY =
issue stall (dependency / pipe busy), SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and sum to ~SQ_WAVE_CYCLES.
usage: python profiles/stall_table.py <sq.csv> <tcp.csv or ''> "<command>" > profiles/<tag>_pmc_stalls.txt"""
import collections
import csv
import sys


def load(path):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    if not path:
        return rows, launches
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:92]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return rows, launches


sq, nsq = load(sys.argv[1])
tcp, ntcp = load(sys.argv[2] if len(sys.argv) > 2 else "")
tcp_cols = sorted({c for k in tcp for c in tcp[k]})
print("rocprofv3 --kernel-trace --pmc <SQ stall set> | --pmc %s over `%s`; averages per launch of a kernel instantiation." % (" ".join(tcp_cols) or "-", sys.argv[3]))
print("wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt or a barrier); stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls);")
print("act = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; ldsst = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES);")
print("waves = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES (resident waves per busy CU, quad-cycle units cancel); tcpst = TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1 when both exist.\n")
hdr = "%-94s %5s %13s %6s %6s %6s %6s %6s %6s %12s" % ("kernel", "n", "WAVE_CYCLES", "wait", "stall", "act", "ldsst", "mfma", "waves", "INSTS_LDS")
for c in tcp_cols:
    hdr += " %14s" % c.replace("TCP_", "")[:14]
hdr += " %6s" % "tcpst"
print(hdr)
for k in sorted(sq, key=lambda k: -sq[k]["SQ_WAVE_CYCLES"]):
    n = max(1, len(nsq[k]))
    c = sq[k]
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    busy = c["SQ_BUSY_CU_CYCLES"] or 1.0
    line = "%-94s %5d %13.0f %6.2f %6.2f %6.2f %6.2f %6.2f %6.1f %12.0f" % (
        k, n, wc / n, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc,
        c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * busy), 4.0 * wc / busy, c["SQ_INSTS_LDS"] / n)
    t = tcp.get(k, {})
    nt = max(1, len(ntcp.get(k, ())))
    for col in tcp_cols:
        line += " %14.0f" % (t.get(col, 0.0) / nt)
    g = t.get("TCP_GATE_EN1", 0.0)
    line += " %6.2f" % (t.get("TCP_PENDING_STALL_CYCLES", 0.0) / g if g else float("nan"))
    print(line)
