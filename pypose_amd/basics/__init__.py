from .ops import pm, cumops, cummul, cumprod, cumops_, cummul_, cumprod_
