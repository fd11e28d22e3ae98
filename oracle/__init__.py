"""oracle/ -- CPU restatements of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the reported CPU baseline.  The product (rotate-yolov3_amd/) never does.
"""
