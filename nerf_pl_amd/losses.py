"""`losses.py` of the reference (losses.py:4-16): MSE on coarse (+ fine) rgb.  Tiny; stays in torch."""
from torch import nn


class MSELoss(nn.Module):
    def __init__(self):
        super(MSELoss, self).__init__()
        self.loss = nn.MSELoss(reduction='mean')

    def forward(self, inputs, targets):
        loss = self.loss(inputs['rgb_coarse'], targets)
        if 'rgb_fine' in inputs:
            loss = loss + self.loss(inputs['rgb_fine'], targets)
        return loss


loss_dict = {'mse': MSELoss}
