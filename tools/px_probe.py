"""px kernel phase probe: MESM_PX_DEBUG=0|1|2|3 (1 = no refill loads, 2 = no matrix instructions) on a few shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import px_check as P
for sh in [(4800, 256, 1024, False, True, 1), (4800, 1024, 256, False, True, 1), (4800, 256, 256, False, True, 1), (1024, 256, 4800, True, False, 4)]:
    for tile in (64, 96):
        e, r, t, ts = P.run(*sh, True, tile)
        print(sh, "tile", tile, "dbg", os.environ.get("MESM_PX_DEBUG", "0"), "%.2f us (min %.2f)" % (t, P.timed.spread[0]), "err %.1e" % e, flush=True)
