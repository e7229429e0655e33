"""HIP backend: ctypes binding of libe3k.so + autograd glue."""
