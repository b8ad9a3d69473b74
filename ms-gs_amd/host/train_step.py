"""One MS-GS training iteration composed from the GPU pieces of this package — the body of the reference's loop
(/root/reference/train.py:202-218 render + loss + backward, :239-250 statistics, :416-418 optimizer step) with every
per-Gaussian / per-pixel pass inside libmsgs_hip.so:

    render_fused        activations + rasterizer forward            (gaussian_renderer.render_fused)
    l1_ssim_loss        photometric loss                            (loss_utils.l1_ssim_loss)
    backward            loss gradient -> rasterizer backward -> raw-parameter gradients
    update_training_stats, FusedAdam.step                           (train_epilogue)

The densification POLICY (when to clone / split / prune / insert, train.py:252-262) stays with the caller: it is
control plane, out of scope here.
"""
import torch

import diff_gaussian_rasterization as dgr
from gaussian_renderer import render_fused
from loss_utils import l1_ssim_loss
from train_epilogue import update_training_stats


def fused_train_iteration(model, optimizer, cam, gt_image, pipe, bg, *, lambda_dssim=0.2, loss_multiplier=1.0,
                          reso_lvl=0, filter_small=False, filter_large=False, fade_size=1.0, base_mask=False,
                          update_pixel_sizes=True, densify=True, step_in_backward=False):
    """Returns (loss, Ll1, render_pkg); loss / Ll1 are 0-dim GPU tensors (no host sync is issued here).
    step_in_backward: the Adam step of the six leaf tensors is taken INSIDE the rasterizer's per-Gaussian backward kernel
    (include/msgs.h, msgs_adam_in_backward_t): no gradient tensors are written or read (2 x 236 bytes per Gaussian), the
    parameters and moments come out bit-identical to the default composition.  The statistics below read only the render
    outputs and the screen-space gradient, so their order against the step does not matter."""
    if step_in_backward:
        taken = getattr(optimizer, "steps_in_backward", 0)
        prev = dgr.set_optimizer_in_backward(optimizer)
        try:
            pkg = render_fused(cam, model, pipe, bg, filter_small=filter_small, filter_large=filter_large, fade_size=fade_size)
        finally:
            dgr.set_optimizer_in_backward(prev)
    else:
        pkg = render_fused(cam, model, pipe, bg, filter_small=filter_small, filter_large=filter_large, fade_size=fade_size)
    loss, Ll1 = l1_ssim_loss(pkg["render"], gt_image, lambda_dssim)
    if loss_multiplier != 1.0:                      # train.py:212-215 (0.1 on the coarser levels)
        loss = loss * loss_multiplier
    # one device, one graph: run the backward on THIS thread instead of handing it to autograd's device thread (the
    # hand-off costs 0.1 ms per iteration on small scenes and, on a busy host, an occasional 4 ms wake-up: hostinfo.py)
    with torch.autograd.set_multithreading_enabled(False):
        loss.backward()
    with torch.no_grad():
        update_training_stats(model, pkg["viewspace_points"], pkg["radii"], pkg["pixel_sizes"], reso_lvl,
                              base_mask=base_mask, update_pixel_sizes=update_pixel_sizes, densify=densify)
        if step_in_backward:
            if getattr(optimizer, "steps_in_backward", 0) != taken + 1:
                raise RuntimeError("fused_train_iteration: the backward did not take the optimizer step (render_fused did not go "
                                   "through the raw rasterizer entry?)")
        else:
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss.detach(), Ll1, pkg


def fused_train_iteration_views(model, optimizer, pipeline, cams, gt_images, pipe, bg, *, lambda_dssim=0.2, loss_multiplier=1.0,
                                reso_lvl=0, filter_small=False, filter_large=False, fade_size=1.0, base_mask=False,
                                update_pixel_sizes=True, densify=True, render_fn=render_fused):
    """ONE optimizer step over SEVERAL views (gradient accumulation over the views of the step, as the view-parallel config C4
    does across GPUs — the reference itself steps after every view, train.py:193-216): the views run through
    multi_view.ViewPipeline two at a time, their gradients are summed inside the per-Gaussian backward kernel, the per-view
    statistics (train.py:239-250) are applied afterwards in view order — update_pixel_sizes is order dependent — and the optimizer
    steps once.  Returns (losses [n] tensor, render packages); same arithmetic as the serial composition (same views, same
    order), bit for bit (tests/test_train_step_gpu.py)."""
    losses = [None] * len(cams)

    def backward_fn(i, pkg):
        loss, _ = l1_ssim_loss(pkg["render"], gt_images[i], lambda_dssim)
        if loss_multiplier != 1.0:
            loss = loss * loss_multiplier
        with torch.autograd.set_multithreading_enabled(False):
            loss.backward()
        losses[i] = loss.detach()
        return pkg, losses[i]         # everything returned here is handed over to the caller's stream (record_stream)
    pkgs = pipeline.train_views(cams, model, pipe, bg, backward_fn, render_fn=render_fn, share_getters=render_fn is not render_fused,
                                filter_small=filter_small, filter_large=filter_large, fade_size=fade_size)
    pkgs = [p_[0] for p_ in pkgs]
    with torch.no_grad():
        for pkg in pkgs:                              # on the caller's stream, after the lanes have joined: view order
            update_training_stats(model, pkg["viewspace_points"], pkg["radii"], pkg["pixel_sizes"], reso_lvl,
                                  base_mask=base_mask, update_pixel_sizes=update_pixel_sizes, densify=densify)
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return torch.stack(losses), pkgs
