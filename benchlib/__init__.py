"""bench.py's parts: common | probe | cpu | single | sharded | models | launch."""
