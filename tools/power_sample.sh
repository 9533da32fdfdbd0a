# Samples package power and shader clock (rocm-smi, every 0.5 s) while bench.py runs the render (cfg5) and the train step (cfg2).
# GPU box: bash tools/power_sample.sh   -> gpurun_out/ab/pw_cfg{2,5}.{txt,json}; summary in profiles/r02_power_clock_samples.json
mkdir -p gpurun_out/ab
python -c "import torch" 2>/dev/null
for cfg in cfg5 cfg2; do
  st=400; [ $cfg = cfg2 ] && st=250
  ( python bench.py --config $cfg --steps $st --warmup 5 --no-cpu-baseline > gpurun_out/ab/pw_$cfg.json 2>/dev/null ) &
  BP=$!
  : > gpurun_out/ab/pw_$cfg.txt
  while kill -0 $BP 2>/dev/null; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*: //' | tr '\n' ' ' >> gpurun_out/ab/pw_$cfg.txt; echo >> gpurun_out/ab/pw_$cfg.txt; sleep 0.5; done
  echo "== $cfg"; sort -t'(' -k2 -n gpurun_out/ab/pw_$cfg.txt | awk 'NF' | tail -8
done
