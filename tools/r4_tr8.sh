#!/bin/bash
# round 4: the transposing compress with 256-column tiles (k_compress_tr8w) against the 64-column tiles
for w in 0 128 256; do
  TD_TR8_WIDE=$w timeout 300 python bench.py --workload g3 --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide=$w', round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, d['total_cost'])"
done
for w in 128 256; do
TD_TR8_WIDE=$w timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_forest.py "tests/test_gpu_parity.py::test_rectangular_models_constant_rows_and_columns" "tests/test_gpu_parity.py::test_padded_model_whose_real_cells_exceed_the_pad_value" -m gpu -x -q 2>&1 | tail -2
done
