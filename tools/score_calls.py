import csv, sys
rows=list(csv.DictReader(open("gpurun_out/r02_h_score/trace_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sc=[r for r in rows if "score_" in r["Kernel_Name"] or "pack_items" in r["Kernel_Name"]]
calls=[]; cur=[]
for r in sc:
    if "pack_items" in r["Kernel_Name"] and cur:
        calls.append(cur); cur=[]
    cur.append(r)
calls.append(cur)
def show(c):
    t0=int(c[0]["Start_Timestamp"])
    for r in c:
        n=r["Kernel_Name"].split("(")[0].replace("void chaorec::","").replace("chaorec::","")
        print("   %8.1f +%7.1f us  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, n[:60]))
print(len(calls),"calls")
for i in [int(a) for a in sys.argv[1:]] or (3, 10, len(calls)-2):
    print("call", i); show(calls[i])
