timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_ddp.py tests/test_gpu_function.py -x -q 2>&1 | tail -4
UDP_POSE_NO_CONV_MULTI=1 python tools/bench_train.py --dtype bf16 --steps 10 --warmup 3 2>&1 | tail -1
python tools/bench_train.py --dtype bf16 --steps 10 --warmup 3 2>&1 | tail -1
python tools/bench_train.py --dtype f32 --steps 10 --warmup 3 2>&1 | tail -1
