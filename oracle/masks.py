"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's tracking-mask -> ``routing_logits_forcing`` path
(stage 2 of its inference): reference util/utils.py:481-514 (``resize_mask``) and :871-936
(``process_masks_to_routing_logits``, parts 2-3; part 1, reading PNG frames, is host I/O and stays with the caller).

PINNED: ``tests/golden/ref_masks_seed*.npz`` holds the output of the reference functions themselves (imported
unmodified by ``tests/golden/make_golden.py --case masks``) on the seeded synthetic masks of
``tests/golden/mask_cases.py``; ``tests/test_oracle_golden.py`` checks this file against them bit for bit.
"""
import torch
import torch.nn.functional as F


def resize_mask(mask, size):
    """reference util/utils.py:505-512 (``process_first_frame_only=False``): trilinear, align_corners=False."""
    return F.interpolate(mask, size=list(size), mode="trilinear", align_corners=False)


def masks_to_routing_logits(masks, latent_frames=13, height=60, width=90, patch=2):
    """masks: [n_id, T, H, W] (anything > 0 is foreground, reference util/utils.py:857) -> [1, T'*h*w, n_id] float.
    Later identities overwrite earlier ones where masks overlap (reference :912-913); background rows are all zero."""
    n_id = masks.shape[0]
    size = (latent_frames, height // patch, width // patch)
    index = torch.full((1, 1) + size, -1, dtype=torch.long)
    for i in range(n_id):
        cur = (masks[i] > 0).to(torch.uint8).unsqueeze(0).float().unsqueeze(1)          # [1, 1, T, H, W]
        binary = (resize_mask(cur, size) > 0.5).long()
        index = torch.where(binary == 1, torch.tensor(i, dtype=torch.long), index)
    index = index.reshape(1, -1)
    logits = torch.zeros(1, index.shape[1], n_id)
    for i in range(n_id):
        logits[0, index[0] == i, i] = 1
    return logits
