import csv,glob,collections,sys,re
fs=glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k=re.sub(r"\(.*$","",r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::",""))[:50]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("GRBM_GUI_ACTIVE",0)):
    w=v.get("SQ_WAVES",1); cyc=v.get("GRBM_GUI_ACTIVE",0)/8
    if cyc<=0 or n[k]==0: continue
    print(f"{k:50s} launches {n[k]:3d} waves/launch {w/n[k]:9.0f} valu/wave {v['SQ_INSTS_VALU']/w:7.1f} salu/wave {v.get('SQ_INSTS_SALU',0)/w:7.1f} wave_cyc {v['SQ_WAVE_CYCLES']/w:8.1f} valu_util {v['SQ_ACTIVE_INST_VALU']*4/(1024*cyc):.3f} waves_in_flight/SIMD {v['SQ_WAVE_CYCLES']*4/(1024*cyc):.2f} us {cyc/n[k]/2100:.1f}")
