"""regular package so that it wins over an unrelated site-packages `datasets`"""
