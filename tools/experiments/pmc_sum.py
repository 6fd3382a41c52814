import csv,glob,collections,sys
out=sys.argv[1]
for pas in ('p1','p2','p3'):
    fs=glob.glob(f'{out}/{pas}/**/*counter_collection.csv',recursive=True)
    if not fs: print(pas,'no file'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        n=r['Kernel_Name'].split('(')[0][-60:]
        agg[n][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']==list(agg[n].keys())[0]: cnt[n]+=1
    for n,d in agg.items():
        if 'gemm_nt' in n: print(pas,n,cnt[n],{k:round(v/cnt[n]/1e6,3) for k,v in d.items()})
