import csv,glob,os,sys
fs=sorted(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows=[r for r in csv.DictReader(open(fs[-1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:50]) for r in rows]
# find runs of grid_pass batch kernels
out=[]
prev=None
for s,e,k in seq:
    if "grid_pass_kernel" in k or "gridb_pass_kernel" in k:
        if prev is not None: out.append(((s-prev)/1e3,(e-s)/1e3))
        prev=e
    else:
        if prev is not None and "streamOps" not in k: pass
print("n",len(out))
import statistics
g=[x for x,_ in out if x<500]
print("gap median %.1f mean %.1f  kernel median %.1f"%(statistics.median(g), sum(g)/len(g), statistics.median([d for _,d in out])))
print(" ".join("%.0f/%.0f"%(x,d) for x,d in out[:50]))
