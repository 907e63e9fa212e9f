F='amdgpu.ids\|Gloo\|socket.cpp'
export DIAG_ONECALL=1 DIAG_NOSYNC=1
for rep in 1 2; do
echo "== V0 as is"; python tools/diag_shared_gpu.py 4 2>&1 | grep -v "$F"
echo "== V1 no overlap"; MORL_COMM_NO_OVERLAP=1 python tools/diag_shared_gpu.py 4 2>&1 | grep -v "$F" | grep -v unsharded
echo "== V2 device sync around collectives"; MORL_COMM_DEBUG=sync python tools/diag_shared_gpu.py 4 2>&1 | grep -v "$F" | grep -v unsharded
echo "== V3 host-staged all-reduce"; MORL_COMM_DEBUG=hostar python tools/diag_shared_gpu.py 4 2>&1 | grep -v "$F" | grep -v unsharded
done
