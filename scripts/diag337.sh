mkdir -p gpurun_out/r06e
for lib in libjuqbox_hip.so libjuqbox_hip_df.so; do
  echo "== $lib oracle mode"
  JQ_LIB=$PWD/juqbox.jl_amd/$lib FUZZ_FOCUS=slab FUZZ_ONLY=337 python scripts/fuzz_gpu_r06d.py 350 6102 2>&1 | grep -v "weights:" | tail -4
  JQ_LIB=$PWD/juqbox.jl_amd/$lib FUZZ_FOCUS=slab FUZZ_ONLY=337 FUZZ_DUMP=gpurun_out/r06e/d337_$lib.json python scripts/fuzz_gpu_r06d.py 350 6102 > /dev/null 2>&1
done
python - <<'P'
import json
a=json.load(open("gpurun_out/r06e/d337_libjuqbox_hip.so.json")); b=json.load(open("gpurun_out/r06e/d337_libjuqbox_hip_df.so.json"))
print(a.keys(), b.keys())
for k in a:
    print(k, a[k]==b[k])
    if a[k]!=b[k]:
        print(" main:", str(a[k])[:600]); print(" df:  ", str(b[k])[:600])
P
