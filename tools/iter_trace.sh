cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=gpurun_out/it1; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/tr -o wf -- python3 bench.py --no-cpu --steps 1 --warmup 0 --lanes 1 > $out/log 2>&1 || exit 1
python3 tools/iter_times.py $(find $out/tr -name '*.db' | head -1) 8 > $out/iters.txt
rm -rf $out/tr
