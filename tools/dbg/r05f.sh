mkdir -p gpurun_out/r05
bash tools/trace_step.sh r05
python tools/nomfma_gaps.py gpurun_out/r05/bench_trace.db 40 > gpurun_out/r05/r05_nomfma_gaps.txt 2>&1
head -3 gpurun_out/r05/r05_nomfma_gaps.txt
bash tools/roofline_table.sh r05
tail -5 gpurun_out/r05/r05_roofline_table.txt
python -m pytest tests/test_model_gpu.py -q -m gpu -k "direct_form_second or c4_full_batch or local_enhancer_full or c2_teacher_forced or c1_teacher_forced_20 or two_stream_encoder or tiny_twostream_teacher or tiny_global_teacher" --durations=12 > gpurun_out/r05/parity_rerun.log 2>&1; tail -22 gpurun_out/r05/parity_rerun.log
