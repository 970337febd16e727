import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# find replays by adam_kernel
steps=[];cur=[]
for r in rows:
    cur.append(r)
    if 'adam_kernel' in r["Kernel_Name"]:
        steps.append(cur);cur=[]
print("replays", len(steps))
for i in [30, 250, 330, 500, 700, 880]:
    if i < len(steps):
        s=steps[i]
        t0=int(s[0]["Start_Timestamp"]);t1=int(s[-1]["End_Timestamp"])
        busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in s)
        print(i,len(s),"span %.0f us busy %.0f us"%((t1-t0)*1e-3,busy*1e-3))
        if i in (330, 700):
            prev=None
            for r in s:
                st,en=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
                print("   gap %7.1f  dur %7.1f  %s"%(((st-prev)*1e-3 if prev else 0),(en-st)*1e-3,r["Kernel_Name"][:50]))
                prev=en
