"""Summarises the per-kernel PMC counters collected by scripts/pmc.sh (clock, MFMA-busy fraction, LDS conflicts, L2 hit, traffic)."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1]
# --json "<bench.py roofline.kernel label>" "<kernel name substring>": one record for bench.py's roofline.traffic
want_json = sys.argv[3:5] if len(sys.argv) >= 5 and sys.argv[2] == "--json" else None
records = []
agg = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/{tag}/p*/*/*counter_collection.csv")):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'conv' not in name: continue
        k = (name[:64], r['Grid_Size'], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
        d = agg.setdefault(k, collections.defaultdict(list))
        d[r['Counter_Name']].append(float(r['Counter_Value']))
        d['_dur'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    dur = m['_dur']
    line = f"{k[0]:64s} grid={k[1]:>9s} lds={k[2]:>6s} vgpr={k[3]}+{k[4]} dur={dur/1e3:8.1f}us"
    if 'SQ_LDS_IDX_ACTIVE' in m:
        line += f" ldsconf={m['SQ_LDS_BANK_CONFLICT']/max(m['SQ_LDS_IDX_ACTIVE'],1):.2f} mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES']/m['SQ_BUSY_CYCLES']:.1f} wait_any={m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.2f} wait_inst={m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.2f}"
    if 'TCC_HIT_sum' in m:
        line += f" L2hit={m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.2f}"
    if 'FETCH_SIZE' in m:
        line += f" fetchMB(x2)={2*m['FETCH_SIZE']/1024:.0f}"
    if 'WRITE_SIZE' in m:
        line += f" writeMB={m['WRITE_SIZE']/1024:.0f}"
    if 'SQ_INSTS_MFMA' in m:
        line += f" valu/mfma={m['SQ_INSTS_VALU']/max(m['SQ_INSTS_MFMA'],1):.2f} lds/mfma={m['SQ_INSTS_LDS']/max(m['SQ_INSTS_MFMA'],1):.2f} vmemrd/mfma={m['SQ_INSTS_VMEM_RD']/max(m['SQ_INSTS_MFMA'],1):.3f} wait_lds={m.get('SQ_WAIT_INST_LDS',0)/max(m.get('SQ_ACTIVE_INST_ANY',1),1):.2f}"
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        # rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; MFMA busy cycles are summed over all 1024 SIMDs
        clk = m['GRBM_GUI_ACTIVE'] / 8 / dur  # GHz (dur in ns)
        line += f" clk={clk:.2f}GHz mfma_util={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}"
    records.append((k[0], m, dur))
    if not want_json:
        print(line)
if want_json:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    label, sub = want_json
    name, m, dur = next(r for r in records if sub in r[0])
    rec = {"kernel_label": label, "kernel_name": name, "source_sha16": bench.kernel_source_sha(),
           "fetch_MB": round(2 * m['FETCH_SIZE'] / 1024, 1), "write_MB": round(m['WRITE_SIZE'] / 1024, 1), "dur_us": round(dur / 1e3, 1),
           "L2hit": round(m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 3),
           "mfma_util": round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024), 3),
           "clk_GHz": round(m['GRBM_GUI_ACTIVE'] / 8 / dur, 3),
           "note": "rocprofv3 --pmc, separate passes (scripts/pmc.sh); FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md); MiB per launch"}
    print(json.dumps(rec))
