O=gpurun_out/r04g; mkdir -p $O
for i in 11 12 15 16; do timeout 120 python scripts/conv_stamps.py $i 2>&1 | grep -v amdgpu.ids | awk 'NR==1 || /step loop/ && ++c<=3 || /barrier/ && ++b<=3 || /load\+transform|index math|epilogue|total|distinct|event|span/' ; done > $O/stamps_small.txt
cat $O/stamps_small.txt
