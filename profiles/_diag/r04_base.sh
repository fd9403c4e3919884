mkdir -p gpurun_out/r04_base
for W in c2 c3 c4 c5; do
  python bench.py --workload $W --no-cpu-baseline --repeats 3 --no-cold-pass > gpurun_out/r04_base/$W.json 2> gpurun_out/r04_base/$W.err
done
python bench.py --workload c2 --envs 512 --no-cpu-baseline --repeats 3 --no-cold-pass > gpurun_out/r04_base/c2_512.json 2> gpurun_out/r04_base/c2_512.err
python bench.py --workload c2 --envs 1024 --no-cpu-baseline --repeats 3 --no-cold-pass > gpurun_out/r04_base/c2_1024.json 2> gpurun_out/r04_base/c2_1024.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_base/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.2f M"%(d["value"]/1e6), "ms/step %.4f"%d["ms_per_step"], "kernel_ms %.4f"%d["roofline"]["kernel_ms"], "frac %.3f"%d["roofline"]["frac"], [round(v/1e6,2) for v in d["repeats"]["values"]])
    except Exception as e:
        print(f, "ERR", e)
PY
