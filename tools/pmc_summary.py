import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
pat=sys.argv[2] if len(sys.argv)>2 else ''
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in rows:
    k=r['Kernel_Name'].split('(')[0][:120]  # template arguments included, parameter list dropped
    if pat and pat not in k: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k,v in agg.items():
    print(k, "dispatches", len(disp[k]))
    for c,val in sorted(v.items()): print("    %-32s %.4g  (per dispatch %.4g)"%(c,val,val/len(disp[k])))
