set -e
B="--train-steps 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
for cfg in "8 1" "8 2" "8 4" "16 2" "16 4" "32 4"; do set -- $cfg
  timeout -k 10 200 python3 bench.py --graphs $1 --parts $2 $B > gpurun_out/sb_$1_$2.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/sb_$1_$2.json'));print('graphs $1 parts $2', round(d['value'],1), round(d['ms_per_step'],1))"
done
