run() { echo "== $*"; env "$@" python tools/multiqueue_conv.py --tracks 1024 --buffers 4000 2>&1 | grep '"host_threads": [123],\|"ranges": 1,'; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
