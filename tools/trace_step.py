import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Dispatch_Id']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_t_adam')]
a,b=idx[-2]+1,idx[-1]+1
tot=0
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    name=r['Kernel_Name'].split('(')[0][:34]
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    grid=(int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),int(r['Grid_Size_Y'])//max(int(r['Workgroup_Size_Y']),1))
    tot+=dur
    if dur>float(sys.argv[2]): print(f"{name:36s} grid={str(grid):12s} start={(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur={dur:9.1f} us  stream={r.get('Stream_Id','')}")
print('total',tot, 'span', (int(rows[b-1]['End_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3)
