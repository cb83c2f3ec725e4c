set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
# 1. the JNI shim harness, its own output
g++ -O1 -std=c++17 -Itests/native/jni_stub -Iinclude tests/native/jni_shim_test.cpp -o /tmp/jni_shim_test -Lembedding_amd -l:libdge.so -Wl,-rpath,$PWD/embedding_amd -Wl,-rpath,/opt/rocm/lib && /tmp/jni_shim_test /tmp > gpurun_out/r05_jni_shim_test.txt 2>&1
echo "jni rc=$?"; cat gpurun_out/r05_jni_shim_test.txt
# 2. cfg2: one mini-batch a launch (default) against two / three (the generator and the first sort of the next beside the phases of this one)
f() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  %.4e edges/s  %.3f ms/step  %.3f ms/launch' % (d['value'], d['ms_per_step'], r['ms_per_launch']), flush=True)"; }
for sw in 0 50000 33334 0; do
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --workload cfg2 --tune sorted_walks=$sw 2>/dev/null | f "cfg2 sorted_walks=$sw" || exit 1
done > gpurun_out/r05_cfg2_minibatches_per_launch.txt
cat gpurun_out/r05_cfg2_minibatches_per_launch.txt
# 3. the epoch line with the CPU projection
timeout -k 10 400 python3 bench.py --epoch > gpurun_out/r05_epoch.json 2> gpurun_out/r05_epoch.err || exit 1
tail -c 1500 gpurun_out/r05_epoch.json
