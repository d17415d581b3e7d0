)fX)ژ'	i[քxU6v0l_lCvqv
̺ilǡȵr=.4yMU&[-˙1QLV[WШ&ч`皽ɸlybH]Y5LiMxU+2@J)f7LA,k9\%$)̚lq;κ4R
Ti;[ I1/WNK=