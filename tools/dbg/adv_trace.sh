mkdir -p gpurun_out/adv; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/adv/t -- python3 bench.py --steps 2 --warmup 1 --end-to-end 0 --traffic 0 --cpu-seconds 0 --check-pages 0 > gpurun_out/adv/bench.json 2> gpurun_out/adv/bench.err
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/adv/t/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=open('gpurun_out/adv/last_kernels.txt','w')
for r in rows[-60:]:
    out.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:12.1f} us  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r['Grid_Size_X']:>9} wg {r['Workgroup_Size_X']:>5}  {r['Kernel_Name'][:90]}\n")
PY
tail -45 gpurun_out/adv/last_kernels.txt
