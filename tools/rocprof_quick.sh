cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp -o x -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2> gpurun_out/rp.log
python3 - <<'PY'
import pandas as pd, glob
d=pd.read_csv(glob.glob('gpurun_out/rp/**/x_kernel_stats.csv', recursive=True)[0])
d['Name']=d['Name'].str.replace(r'\(.*','',regex=True).str.slice(0,60)
print(d[['Name','Calls','AverageNs','MaxNs','Percentage']].head(14).to_string())
PY
rm -rf gpurun_out/rp
