"""Merge launch-table rows (tools/patch_autotune.py / step_autotune.py / conv_autotune.py --out) into csrc/conv_tune_table.inc: a new row
replaces the row with the same key (kind, M, Ca, Cb, Ktot | KH, pc); everything else is kept.    python tools/merge_tune_rows.py rows.inc"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "optimalstrategiesagainstgenerativeattacks_amd", "csrc", "conv_tune_table.inc")
ROW = re.compile(r"\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), \d+, \d+\}")
new = [ln.rstrip("\n") for ln in open(sys.argv[1]) if ROW.search(ln)]
keys = {ROW.search(ln).groups() for ln in new}
kept, dropped = [], 0
for ln in open(TABLE):
    mt = ROW.search(ln)
    if mt and mt.groups() in keys:
        dropped += 1
        continue
    kept.append(ln.rstrip("\n"))
open(TABLE, "w").write("\n".join(kept + ["// merged from %s" % os.path.basename(sys.argv[1])] + new) + "\n")
print("replaced %d rows, added %d" % (dropped, len(new) - dropped))
