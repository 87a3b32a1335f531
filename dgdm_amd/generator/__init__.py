"""Reference-shaped ``generator`` package (names of /root/reference/generator) backed by libdgdm_hip.so."""
