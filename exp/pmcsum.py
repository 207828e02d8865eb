import csv,glob,collections,sys
for name in sys.argv[1:]:
    f=glob.glob('/root/repo/gpurun_out/%s/*/*counter_collection.csv'%name)[0]
    rows=list(csv.DictReader(open(f)))
    agg=collections.OrderedDict()
    for r in rows:
        k=r['Kernel_Name'].split('(')[0][-28:]+'#'+r['Dispatch_Id']
        agg.setdefault(k,collections.OrderedDict())
        agg[k][r['Counter_Name']]=agg[k].get(r['Counter_Name'],0)+float(r['Counter_Value'])
        agg[k]['us']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    for k,v in list(agg.items())[-11:]:
        print(k, ' '.join('%s=%.4g'%(a,b) for a,b in v.items()))
