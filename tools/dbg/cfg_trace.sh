# kernel timeline of one bench configuration: tools/dbg/cfg_trace.sh <name> <bench.py args...>   -> gpurun_out/trace/<name>.txt
name=$1; shift
mkdir -p gpurun_out/trace; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace/t_$name
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace/t_$name -- python3 bench.py "$@" --steps 3 --warmup 1 --worst-case 0 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 0 --ceilings 0 > gpurun_out/trace/$name.json 2> gpurun_out/trace/$name.err
python3 - $name <<'PY'
import csv,glob,sys
name=sys.argv[1]
f=glob.glob(f'gpurun_out/trace/t_{name}/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=open(f'gpurun_out/trace/{name}.txt','w')
prev=None
for r in rows[-90:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev)/1e3 if prev else 0.0
    prev=e
    out.write(f"{(s-t0)/1e3:12.1f} us  gap {gap:8.1f}  dur {(e-s)/1e3:9.1f} us  grid {r['Grid_Size_X']:>9} wg {r['Workgroup_Size_X']:>5}  {r['Kernel_Name'][:100]}\n")
PY
rm -rf gpurun_out/trace/t_$name
tail -50 gpurun_out/trace/$name.txt
