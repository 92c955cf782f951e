"""Per-kernel matrix-pipe utilisation and stall shares from one rocprofv3 --pmc pass over the SQ counters
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU).  mfma_util = MFMA_BUSY / (4 * BUSY_CU) as in DESIGN.md section 3."""
import collections, csv, glob, json, sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        n[k] += 1
res = {}
for k, c in agg.items():
    busy = c.get('SQ_BUSY_CU_CYCLES', 0.0)
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    res[k] = dict(launches=n[k], mfma_util=round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (4 * busy), 4) if busy else None,
                  wait_any=round(c.get('SQ_WAIT_ANY', 0.0) / wc, 4) if wc else None,
                  wait_inst_any=round(c.get('SQ_WAIT_INST_ANY', 0.0) / wc, 4) if wc else None,
                  wait_inst_lds=round(c.get('SQ_WAIT_INST_LDS', 0.0) / wc, 4) if wc else None,
                  lds_bank_conflict_per_wave_cycle=round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / wc, 5) if wc else None,
                  busy_cu_cycles=busy, valu_insts=c.get('SQ_INSTS_VALU', 0.0))
json.dump(res, open(out, 'w'), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]['busy_cu_cycles'])[:40]:
    print(f'{k[:70]:70s} n={v["launches"]:4d} mfma={v["mfma_util"]} wait_any={v["wait_any"]} wait_inst={v["wait_inst_any"]} lds={v["wait_inst_lds"]} bank={v["lds_bank_conflict_per_wave_cycle"]}')
