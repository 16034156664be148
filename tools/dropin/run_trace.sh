cd $GRAFT_REPO_ROOT
python3 tools/dropin/run.py 5 > /dev/null 2>&1
mkdir -p /tmp/tl && cp folve_amd/libfolve_amd_trace.so /tmp/tl/libfolve_amd.so
g++ -O2 -std=c++17 -pthread -Iinclude tools/dropin/dropin_threads.cpp -o /tmp/dropin_threads_trace -L/tmp/tl -lfolve_amd -ldl -Wl,-rpath,/tmp/tl
for nt in 1 16 64; do /tmp/dropin_threads_trace /tmp/dropin_cfg3/filter-44100.conf $nt 400 1 2>&1 | grep -v amdgpu; done
