import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last pass: find last k_extract and the previous one
ex=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_extract')]
a,b=ex[-2]+1,ex[-1]
seg=rows[a:b]
iv=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Queue_Id'],r['Kernel_Name'][:28]) for r in seg]
span=iv[-1][1]-iv[0][0]
busy=0; cs,ce=iv[0][:2]
for s,e,_,_ in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print(len(iv),'span ms',span/1e6,'union busy',busy/1e6,'sum',sum(e-s for s,e,_,_ in iv)/1e6)
from collections import defaultdict
acc=defaultdict(lambda:[0,0.0])
for s,e,q,nm in iv: acc[nm][0]+=1; acc[nm][1]+=(e-s)/1e3
for k,v in sorted(acc.items(),key=lambda kv:-kv[1][1])[:8]: print(k, v[0], round(v[1]/v[0],1),'us', round(v[1]/1e3,1),'ms')
