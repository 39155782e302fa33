#!/bin/bash
# tools/sweep_models.sh OUT.json — bench.py (tg128 + pp512, no CPU baseline) over the model shapes / formats of SURVEY.md Appendix B
out=${1:-gpurun_out/sweep.json}
echo "[" > $out; first=1
run() { line=$(python bench.py --no-cpu-baseline --no-profile "$@" 2>/dev/null | tail -1); [ -n "$line" ] || line="{\"failed\": \"$*\"}"; [ $first = 1 ] || echo "," >> $out; first=0; echo "$line" >> $out; echo "$* -> $(echo $line | cut -c1-140)"; }
run --model llama3-8b --ftype Q4_K_M
run --model llama3-8b --ftype Q4_0
run --model llama3-8b --ftype Q6_K
run --model llama3-8b --ftype Q8_0
run --model llama3-8b --ftype Q4_K_M --fa 1
run --model llama3-8b --ftype Q4_K_M --fa 1 --ctk q8_0 --ctv q8_0
run --model mixtral-8x7b --ftype Q4_K_M
run --model gpt-oss-20b --ftype MXFP4_MOE
run --model llama3-70b --ftype Q4_K_M --steps 64 --warmup 16
run --model llama3-8b --ftype Q4_K_M --steps 1024 --warmup 128 --pp 2048
echo "]" >> $out
