"""Run tests/test_gpu_blocks_scale.py's skewed case with trainer knobs set for the whole process (dge_set_tuning is process-wide), to read what a knob does to
the test's own AUC / loss print-out:   python scripts/run_test_with_knobs.py acc_drain=2     (comma-separated knob=value pairs; "" = none).
profiles/README.md quotes its output for the accumulator banks (0.9411 without, 0.9394 / 0.9386 / 0.9377 at 2 / 3 / 4 updates a flush)."""
import sys
sys.path.insert(0, '.')
import embedding_amd as E, pytest
knobs = dict(kv.split('=') for kv in sys.argv[1].split(',')) if len(sys.argv) > 1 and sys.argv[1] else {}
with E.tuning(**{k: int(v) for k, v in knobs.items()}):
    sys.exit(pytest.main(["tests/test_gpu_blocks_scale.py", "-q", "-s", "-k", "community_zipf"]))
