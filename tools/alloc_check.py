"""Run the default bench.py and print the caching allocator's device alloc / free counters at every synchronize: the timed region
must not allocate device memory (it does not: the counters stop moving after the warm-up)."""
import sys, json, runpy, torch
sys.argv = ['bench.py', '--no-cpu-baseline']
sys.path.insert(0, "."); import bench
orig_sync = torch.cuda.synchronize
stats = []
def hook(*a, **k):
    s = torch.cuda.memory_stats()
    stats.append((s.get('num_device_alloc', -1), s.get('num_device_free', -1), s.get('num_alloc_retries', -1), s.get('reserved_bytes.all.current', 0) / 2**30))
    return orig_sync(*a, **k)
torch.cuda.synchronize = hook
bench.main()
print('device allocs / frees / retries / reserved GiB at every synchronize:', file=sys.stderr)
for s in stats: print('  ', s, file=sys.stderr)
