cp readsb_amd/libreadsb_amd.so /tmp/keep.so
cp variants/clocks.so readsb_amd/libreadsb_amd.so
timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --steps-in-flight 1 2>&1 | grep -E "emit clocks|stage clocks" | tail -3
cp /tmp/keep.so readsb_amd/libreadsb_amd.so
