import sqlite3, collections, sys, glob, json
out={}
for f in glob.glob(sys.argv[1]+'/**/*_results.db', recursive=True):
    db=sqlite3.connect(f); cur=db.cursor()
    rows=list(cur.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels order by start"))
    agg=collections.defaultdict(list)
    meta={}
    for r in rows:
        k=r[0].split('(')[0].replace('void ',''); agg[k].append(r[2]-r[1]); meta[k]=r[3:]
    tot=sum(sum(v) for v in agg.values())
    print("%-60s %7s %10s %10s %10s %10s %6s  grid/wg/vgpr/agpr/sgpr/lds/scratch"%("kernel","calls","total_us","avg_us","med_us","min_us","pct"))
    for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
        v2=sorted(v)
        print("%-60s %7d %10.1f %10.2f %10.2f %10.2f %6.1f  %s"%(k[:60],len(v),sum(v)/1e3,sum(v)/len(v)/1e3,v2[len(v2)//2]/1e3,v2[0]/1e3,100*sum(v)/tot,meta[k]))
