import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+'/*/*kernel_trace.csv')[0]
    rows=[r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_grid_maintain')]
    x=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
    big=[v for v in x if v>20]; small=[v for v in x if v<=20]
    print(d, 'rebuilds', [round(v) for v in big], 'noop avg', round(sum(small)/len(small),2))
