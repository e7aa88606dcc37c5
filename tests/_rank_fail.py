"""Stand-in worker for the spawn_ranks supervision tests: rank 1 dies (RANK_FAIL_MODE=die) or every rank hangs (hang)."""
import os
import sys
import time

mode = os.environ.get("RANK_FAIL_MODE", "die")
if mode == "die" and os.environ["RANK"] == "1":
    sys.exit(7)
time.sleep(600)          # a rank waiting in a collective for the partner that died / a rendezvous that never completes
