"""Backend classes of the HIP plugin surface (lazy: importing this package never touches the GPU)."""
