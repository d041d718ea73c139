echo "ROCR=$ROCR_VISIBLE_DEVICES HIP=$HIP_VISIBLE_DEVICES CUDA=$CUDA_VISIBLE_DEVICES GPU_DEVICE_ORDINAL=$GPU_DEVICE_ORDINAL"
ls /sys/class/kfd/kfd/topology/nodes/ 2>&1 | head; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n: $(grep -E 'simd_count|location_id|domain' $n/properties 2>&1 | tr '\n' ' ')"; done 2>&1 | head -20
python -c "
import bench; print('count', bench.gpu_count_without_hip()); print('near', bench.gpu_local_cpus())"
python -m pytest tests/test_ddp_gpu.py -m gpu -x -q -s 2>&1 | tail -30
echo ---- config3 then ddp
python -m pytest tests/test_config3_gpu.py tests/test_ddp_gpu.py -m gpu -x -q 2>&1 | tail -30
