# more hardware queues than HIP's default 4: do more sub-batch streams overlap better?
set -e
B="--train-steps 0 --no-cpu-baseline --no-roofline --steps 3 --warmup 1"
for cfg in "4 3" "8 3" "8 4" "8 6" "8 8" "16 8"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python3 bench.py --parts $2 $B > gpurun_out/q_$1_$2.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/q_$1_$2.json'));print('queues $1 parts $2', round(d['value'],1))"
done
