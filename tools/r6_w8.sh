cd /root/repo
MM2C_LIB_PATH=/root/repo/minimap2-fpga_amd/variants/seed_probe.so timeout -k 10 600 python tools/long_reads.py --no-dp --reps 1 --check 0 --sizes 256x1000000 2>&1 | grep "^replay\|^seed" | tail -12
MM2C_LIB_PATH=/root/repo/minimap2-fpga_amd/variants/seed_probe.so timeout -k 10 600 python tools/long_reads.py --no-dp --reps 1 --check 0 --sizes 2048x100000 2>&1 | grep "^replay\|^seed" | tail -8
