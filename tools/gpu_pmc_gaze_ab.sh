ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in ab_base.so libd2d_hip.so; do
  rm -rf $ROOT/gpurun_out/gabl/$lib
  D2D_LIB=$ROOT/gym-drone2d-activeperception_amd/csrc/$lib timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $ROOT/gpurun_out/gabl/$lib -- python3 $ROOT/bench.py --no-persistent --steps 60 --warmup 40 --no-cpu-baseline --leg closed --prologue 200 --workers 0 --envs 1024 > $ROOT/gpurun_out/gabl_$lib.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$ROOT/gpurun_out/gabl/$lib/**/*counter_collection.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_gaze" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$lib:", {c: round(sum(v[-60:])/len(v[-60:])/1024,1) for c,v in d.items()})
PY
done
