"""Per-kernel SQ counter table from rocprofv3 --pmc counter_collection.csv files (tools/pmc_sq_step.sh):
share of wave cycles parked on s_waitcnt / barriers, stalled at issue, issuing; MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES /
(kernel duration x 2.4 GHz x 1024 SIMDs) - the nominal clock: under load the chip runs 1.9 - 2.3 GHz, so the true fraction is
up to a fifth higher."""
import collections
import csv
import re
import sys

for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"],)
        if key not in seen:
            seen.add(key)
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    tot = sum(dur.values())
    print(f"== {path}  (kernel | share of kernel time | parked | issue-stalled (of which LDS) | issuing | MFMA-busy fraction)")
    for k, t in sorted(dur.items(), key=lambda kv: -kv[1])[:22]:
        c = acc[k]
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        print(f"{k[:74]:74s} {t / tot:6.3f} | {c['SQ_WAIT_ANY'] / wc:5.2f} | {c['SQ_WAIT_INST_ANY'] / wc:5.2f} ({c['SQ_WAIT_INST_LDS'] / wc:4.2f}) | "
              f"{c['SQ_ACTIVE_INST_ANY'] / wc:5.2f} | {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (t * 2.4 * 1024):6.3f}")
