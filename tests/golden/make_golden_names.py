"""Names shared by make_golden.py and the tests (kept free of any reference import)."""
OBJ16 = ['rotate', 'rotate_clockwise', 'rotate_counterclockwise', 'shift_up', 'shift_down', 'shift_left', 'shift_right',
         'clockwise_up', 'clockwise_down', 'clockwise_left', 'clockwise_right', 'counterclockwise_up',
         'counterclockwise_down', 'counterclockwise_left', 'counterclockwise_right', 'convergence']
