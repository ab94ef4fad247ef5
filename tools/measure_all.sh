#!/bin/bash
# Measurement sweep on ONE MI355X (run through gpurun); writes JSON lines.
# usage: tools/measure_all.sh OUTFILE
OUT=${1:-gpurun_out/measurements.jsonl}
: > $OUT
run() { echo "# $*" >> $OUT; python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $OUT; }
run --steps 50                                                  # configs[1]: 1 x 3840x2160, fused
run --steps 20 --path unfused                                   # same, materialising path
run --steps 5 --frames 64                                       # configs[2]: 64 x 3840x2160, fused
run --steps 3 --frames 64 --path unfused                        # configs[2], materialising im2col+MFMA path
run --steps 20 --width 7680 --height 4320                       # configs[3] plane on one GPU
run --steps 5 --width 5760 --height 3240 --frames 8             # configs[4] frames (8 of the 512) on one GPU
run --steps 20 --width 576 --height 576                         # configs[0] plane on the GPU
run --steps 5 --path host --frames 32                           # stream of host frames, transfers overlapped
run --steps 5 --path host --width 5760 --height 3240 --frames 8  # configs[4] as the host sees it: 8 of the 512 frames from pageable memory
run --steps 3 --path host --width 5760 --height 3240 --frames 64 # ... 64 of them
run --steps 10 --path host                                      # PCIe-inclusive host-buffer entry point
run --steps 3 --warmup 1 --path surface                         # the reference call surface on host buffers (32 f32 planes over PCIe)
run --steps 10 --path surface-dev                              # the same two call sites with the 32 planes kept on the device (DevicePlane<float>)
run --steps 20 --width 1920 --height 1080                       # a 1080p plane (the line's `two_lanes`: the same planes alternately on two contexts of the GPU)
run --steps 20 --width 1280 --height 720                        # small planes: where two lanes pay (`two_lanes`)
run --steps 20 --width 960 --height 540
run --steps 20 --width 2560 --height 1440
run --steps 20 --mode refbytes --width 1280 --height 720
run --steps 20 --path pipeline                                  # BGR 1080p -> BGR 4K on device (8f rows + conv path)
run --steps 5 --mode exact                                      # bit-exact VALU mode
run --steps 30 --mode refbytes                                  # the reference's bytes: MFMA kernel + exact fix-up of flagged pixels
run --steps 5 --mode refbytes --frames 64                       # ... 64 x 3840x2160
run --steps 20 --mode refbytes --width 1920 --height 1080       # ... a 1080p plane
run --steps 20 --mode refbytes --path pipeline                  # ... BGR 1080p -> BGR 4K
run --steps 30 --mode refbytes16                                # opt-in: the reference's bytes behind the split-f16 kernel
run --steps 5 --mode refbytes16 --frames 64                     # ... 64 x 3840x2160
run --steps 50 --mode split16                                   # opt-in split-f16 mode, 1 x 3840x2160
run --steps 5 --mode split16 --frames 64                        # opt-in split-f16 mode, 64 x 3840x2160
run --steps 20 --mode split16 --path pipeline                   # BGR 1080p -> BGR 4K with the split-f16 conv path
run --steps 5 --mode split16 --path host --frames 32            # host frame stream with the split-f16 conv path
cat $OUT
