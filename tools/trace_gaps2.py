import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[(r['Kernel_Name'].split('(')[0].replace('void ll::','').replace('ll::','')[:40], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for i in range(1,len(ks)):
    gap=(ks[i][1]-ks[i-1][2])/1e3
    if gap>100: continue
    a=agg[ks[i][0]]; a[0]+=1; a[1]+=gap; a[2]+=(ks[i][2]-ks[i][1])/1e3
for k,(c,g,d) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:8]:
    print("%-42s calls %5d  avg gap before %6.2f us  avg dur %6.2f us"%(k,c,g/c,d/c))
