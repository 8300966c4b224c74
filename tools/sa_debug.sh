set -e
cd $GRAFT_REPO_ROOT
cp ygz_slam_amd/libygz_hip.so /tmp/lib_backup.so
make -C ygz_slam_amd/csrc -B EXTRA=-DYGZ_SA_TIMERS -j16 > /dev/null 2>&1
YGZ_SA_DEBUG=1 timeout 100 python tools/stage_bench.py sparse 2>&1 | grep -E "sa-debug|batch" | tail -3
cp /tmp/lib_backup.so ygz_slam_amd/libygz_hip.so
