"""Stand-in worker for test_bench_spawns_its_own_ranks (prints what bench.spawn_ranks put into the environment)."""
import os

print("rank %s of %s local %s master %s" % (os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"], os.environ["MASTER_ADDR"]))
