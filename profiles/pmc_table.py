"""Reduce a rocprofv3 --pmc counter_collection.csv (SQ counters of one eager bench run) into a per-kernel table:
mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   matrix-pipe busy per SIMD
lds  = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES                 LDS array busy
cfl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE              share of LDS cycles lost to bank conflicts
usage: python profiles/pmc_table.py <counter_collection.csv> "<command>" > profiles/<tag>_pmc_counters.txt"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:96]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
cols = ["SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU"]
print("rocprofv3 --pmc %s over `%s`; average per launch of a kernel instantiation." % (" ".join(cols), sys.argv[2]))
print("mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): matrix-pipe busy per SIMD; lds = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES;")
print("cfl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n")
print("%-98s %8s %14s %14s %14s %12s %14s %6s %6s %6s" % ("kernel", "launches", "BUSY_CU_CYC", "MFMA_BUSY_CYC", "LDS_IDX_ACT", "LDS_CONFL", "INSTS_VALU", "mfma", "lds", "cfl"))
for k in sorted(rows, key=lambda k: -rows[k]["SQ_BUSY_CU_CYCLES"]):
    n = max(1, len(launches[k]))
    c = rows[k]
    busy = c["SQ_BUSY_CU_CYCLES"] or 1.0
    print("%-98s %8d %14.0f %14.0f %14.0f %12.0f %14.0f %6.2f %6.2f %6.2f" % (
        k, n, c["SQ_BUSY_CU_CYCLES"] / n, c["SQ_VALU_MFMA_BUSY_CYCLES"] / n, c["SQ_LDS_IDX_ACTIVE"] / n, c["SQ_LDS_BANK_CONFLICT"] / n,
        c["SQ_INSTS_VALU"] / n, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * busy), c["SQ_LDS_IDX_ACTIVE"] / busy,
        c["SQ_LDS_BANK_CONFLICT"] / (c["SQ_LDS_IDX_ACTIVE"] or 1.0)))
