import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
nm=lambda r:r['Kernel_Name'].replace('void ','').split('(')[0].split('<')[0]
gaps={'iter->acc':[], 'acc->iter':[]}
for a,b in zip(rows,rows[1:]):
    if a['Queue_Id']!=b['Queue_Id']: continue
    g=(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3
    if nm(a)=='k_nn_iter' and nm(b)=='k_accumulate_matches': gaps['iter->acc'].append(g)
    if nm(a)=='k_accumulate_matches' and nm(b)=='k_nn_iter': gaps['acc->iter'].append(g)
for k,v in gaps.items():
    v.sort(); print(k, len(v), 'median %.2f us'%v[len(v)//2], 'mean %.2f'%(sum(v)/len(v)))
