"""Per-kernel MFMA and LDS pipe occupancy from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_LDS_IDX_ACTIVE,
SQ_LDS_BANK_CONFLICT, GRBM_GUI_ACTIVE).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md), summed over the SIMDs; GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8.5 x the dispatch's
duration in shader cycles at the bench's clocks), so duration = GUI / 8.  Reported: averages per dispatch, and
mfma_busy = MFMA cycles / (duration x 1024 SIMDs), lds_busy = LDS array cycles / (duration x 256 CUs),
lds_conflict_share = bank-conflict cycles / LDS array cycles.  JSON on stdout."""
import collections, csv, glob, json, sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
out = {}
for k, c in tot.items():
    n = max(cnt[k].values())
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / max(cnt[k]["GRBM_GUI_ACTIVE"], 1)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(cnt[k]["SQ_VALU_MFMA_BUSY_CYCLES"], 1)
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / max(cnt[k]["SQ_LDS_IDX_ACTIVE"], 1)
    bc = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(cnt[k]["SQ_LDS_BANK_CONFLICT"], 1)
    if gui <= 0:
        continue
    dur = gui / 8.0
    out[k] = {"calls": n, "duration_cycles": dur, "mfma_cycles": mf, "lds_cycles": lds, "lds_conflict_cycles": bc,
              "mfma_busy": mf / (dur * 1024.0), "lds_busy": lds / (dur * 256.0), "lds_conflict_share": bc / max(lds, 1.0)}
print(json.dumps(dict(sorted(out.items(), key=lambda kv: -kv[1]["duration_cycles"] * kv[1]["calls"])), indent=1))
