"""Reference-shaped ``dynamics`` package (names of /root/reference/dynamics) backed by libdgdm_hip.so."""
