#!/bin/bash
# ms per frame and kernel split of every BASELINE config (bench.py --config), per-genome kernel on / off
for cfg in cfg1 cfg2 cfg3 cfg4 cfg5; do for rtc in ${RTCS:-1 0}; do
  FLAME_RTC=$rtc python bench.py --config $cfg --steps 6 --warmup 2 --cpu-seconds 0 --preheat-seconds 1.5 --min-timed-frames 24 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_frame']
print('$cfg rtc=$rtc: %.3f ms/frame %.1f Gsamples/s  [iter %.3f accum+flush %.3f filters %.3f]  fuse %d: %.3f ms' % (d['ms_per_step'], d['value']/1e3, k['iter'], k['accum_flush'], k['filters'], d['config']['fuse_short']['fuse'], d['config']['fuse_short']['ms_per_step']))"
done; done
