set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench_r04
O=gpurun_out/bench_r04
python bench.py --steps 600 --warmup 300 > $O/bench_default_600_300.json 2> $O/err1.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_20_5.json 2> $O/err2.log
python bench.py > $O/bench_noflags.json 2> $O/err2b.log
python bench.py --workload config3 --distinct-worlds 512 > $O/bench_config3.json 2> $O/err3.log
python bench.py --workload config4 --distinct-worlds 512 > $O/bench_config4.json 2> $O/err4.log
python bench.py --workload config5 --distinct-worlds 256 > $O/bench_config5.json 2> $O/err5.log
python bench.py --workload config5-step --distinct-worlds 256 --steps 200 --warmup 100 --prologue 100 > $O/bench_config5-step.json 2> $O/err6.log
python bench.py --workload survivability > $O/bench_survivability.json 2> $O/err7.log
D2D_OUT=chainprof.so D2D_EXTRA_FLAGS=-DD2D_CHAIN_PROF bash gym-drone2d-activeperception_amd/csrc/build.sh > /dev/null 2>&1
python tools/chain_prof.py 2>&1 | grep -v amdgpu > $O/chain_prof_config2_4096.txt
B=1 python tools/chain_prof.py 2>&1 | grep -v amdgpu > $O/chain_prof_config2_lone.txt
WORKLOAD=config5 B=4 WORLDS=4 STEPS=100 python tools/chain_prof.py 2>&1 | grep -v amdgpu > $O/chain_prof_config5_lone.txt
rm -f gym-drone2d-activeperception_amd/csrc/chainprof.so
python tools/search_bench.py --envs 1 2>&1 | grep -v amdgpu > $O/search_bench.txt; python tools/search_bench.py --envs 4096 2>&1 | grep -v amdgpu >> $O/search_bench.txt
B=1 python tools/lone_wave.py 2>&1 | grep -v amdgpu > $O/lone_wave.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], '%.4e'%d['value'], 'frac %.4f'%r.get('frac',0), 'traffic', r.get('traffic'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/chain_prof_config5_lone.txt | cut -c1-330
