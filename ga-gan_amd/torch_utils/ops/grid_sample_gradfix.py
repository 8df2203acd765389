"""Bilinear `grid_sample` for 2-D images (zero padding, `align_corners=False`) with gradients of any
order between the image and the output, as the augmentation pipeline needs for R1 regularisation
through a geometric transform (reference: torch_utils/ops/grid_sample_gradfix.py:31-101).

For a fixed grid the op is linear in the image, so two functions that are each other's derivative
cover every order: `_Sample` (image -> samples) and `_Adjoint` (sample gradients -> image gradient,
plus the first-order gradient of the grid).  Both run PyTorch-ROCm's native sampler kernels; as in the
reference there is no second-order gradient with respect to the grid.  `enabled = False` (the
reference's default, training_loop.py:210 switches it on) routes to `torch.nn.functional.grid_sample`.
"""
import torch

enabled = False


def grid_sample(input, grid):
    if enabled:
        return _Sample.apply(input, grid)
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def _check(input, grid):
    if input.ndim != 4 or grid.ndim != 4 or grid.shape[0] != input.shape[0] or grid.shape[3] != 2:
        raise RuntimeError('grid_sample expects input [N, C, H, W] and grid [N, H_out, W_out, 2]')


class _Sample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        _check(input, grid)
        ctx.save_for_backward(input, grid)
        return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        grad_input, grad_grid = _Adjoint.apply(grad_output, input, grid)
        return grad_input, grad_grid


class _Adjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid):
        mask = (ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        # interpolation 0 = bilinear, padding 0 = zeros, align_corners False
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(grad_output, input, grid, 0, 0, False, mask)
        ctx.save_for_backward(grid)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_input, grad2_grid):
        grid, = ctx.saved_tensors
        if ctx.needs_input_grad[2]:
            raise RuntimeError('grid_sample: second-order gradients with respect to the grid are not supported')
        # grad_input is linear in grad_output with the sampling operator's transpose: its derivative is the sampler
        d_grad_output = _Sample.apply(grad2_input, grid) if ctx.needs_input_grad[0] and grad2_input is not None else None
        return d_grad_output, None, None
