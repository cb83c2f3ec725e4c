# what differs between a box that runs the default bench at ~479 ms per launch and one that runs it at ~418?  (profiles/r02_box_drift.txt)
( rocm-smi --showmaxpower --showperflevel --showclocks --showmclkrange --showsclkrange --showpids 2>/dev/null | grep -vE "^=|^$|WARNING" | tr -s ' \t' ' ' ) > gpurun_out/box_info_idle.txt 2>&1
( sleep 14; for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp --showuse --showmemuse 2>/dev/null | grep -E "sclk|fclk|mclk|socclk|Power|Temperature|busy|use" | sed -e 's/GPU\[0\]\s*: //' | tr '\n' ';' | tr -s ' \t' ' '; echo; sleep 1; done ) > gpurun_out/box_info_busy.txt 2>&1 &
python3 bench.py --no-cpu-baseline --steps 12 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.3e edges/s frac %.3f %.1f ms/launch trials %s copy %.0f read %.0f rewrite %.0f' % (d['value'], r['frac'], r['ms_per_launch'], d['config']['placement_trial_ms'], r['box_copy_GBps'], r['row_read_GBps'], r['row_rewrite_GBps']))"
wait
cat gpurun_out/box_info_idle.txt | head -30; tail -3 gpurun_out/box_info_busy.txt
