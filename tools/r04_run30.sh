# round 4, run 30: for the record — host enqueue time per frame against the device's, and the clocks the device reports while the bench loop runs
python tools/host_rate.py 2>/dev/null | tail -2
(python bench.py --no-cpu-baseline --no-target --steps 200000 --warmup 100 --no-long --latency-frames 5 > /dev/null 2>&1 &) ; sleep 6
for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | head -4; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2; sleep 1; done
sleep 8
