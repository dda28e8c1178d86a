for t in 0 200 300 380 450 550; do echo "ticks $t"; GAB_CONV_PHASE_TICKS=$t python tools/multiqueue_conv.py --tracks 1024 --buffers 4000 2>&1 | grep '"ranges": 2, "host_threads": 2'; done
