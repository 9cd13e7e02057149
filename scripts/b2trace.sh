R=$PWD; O=$R/gpurun_out/b2t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-f32 --reps 1 --no-roofline --pair serial --steps 10 --warmup 3"
rocprofv3 --kernel-trace --output-format csv -d $O/b2 -- $B --batch 2 > $O/b2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/c1 -- $B --config cfg1 --steps 3 --warmup 1 > $O/c1.log 2>&1
cd $R
python3 bench.py --batch 2 --no-cpu-baseline --no-f32 --no-roofline --reps 3 > $O/b2_line.json 2>$O/err
python3 bench.py --batch 2 --no-cpu-baseline --no-f32 --no-roofline --reps 3 --pair serial > $O/b2_line_serial.json 2>>$O/err
tail -c 400 $O/b2_line.json; tail -c 300 $O/b2_line_serial.json
