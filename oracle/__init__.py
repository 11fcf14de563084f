"""TEST INFRASTRUCTURE ONLY.  CPU oracle for the QuarkAudio hot path (see DESIGN.md "Oracle").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
