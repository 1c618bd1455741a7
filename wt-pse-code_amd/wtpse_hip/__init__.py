"""wtpse_hip — MI355X (gfx950) engine behind the WT-PSE drop-in modules `algorithms.py` / `shape_networks.py`.

    lib.py    ctypes binding of libwtpse_hip.so (C ABI: include/wtpse_hip.h)
    build.py  hipcc build of csrc/*.hip
    ops.py    host wrappers: allocate outputs, launch on the current HIP stream
    nn.py     parameter containers (reference state_dict names), flat buffers, block forward/backward schedules
    step.py   the training-step harness: counterpart of Trainer.train_epoch's loop body (Trainer.py:766-914)
    dp.py     data-parallel training over RCCL (one process per GPU)
    synth.py  synthetic fundus-shaped batches
"""
__version__ = "0.1.0"
