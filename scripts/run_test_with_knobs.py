import sys
sys.path.insert(0,'/root/repo')
import embedding_amd as E, pytest
knobs = dict(kv.split('=') for kv in sys.argv[1].split(',')) if sys.argv[1] else {}
with E.tuning(**{k:int(v) for k,v in knobs.items()}):
    pytest.main(["tests/test_gpu_blocks_scale.py","-q","-s","-k","community_zipf"])
