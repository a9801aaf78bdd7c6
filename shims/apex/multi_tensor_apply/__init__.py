"""apex.multi_tensor_apply.multi_tensor_applier (apex/multi_tensor_apply/multi_tensor_apply.py): the reference calls
`multi_tensor_applier(op, noop_flag, tensor_lists, *args)` -> `op(chunk_size, noop_flag, tensor_lists, *args)`."""


class MultiTensorApply:
    available = True
    warned = False

    def __init__(self, chunk_size):
        self.chunk_size = chunk_size

    def __call__(self, op, noop_flag_buffer, tensor_lists, *args):
        return op(self.chunk_size, noop_flag_buffer, tensor_lists, *args)


multi_tensor_applier = MultiTensorApply(2048 * 32)
