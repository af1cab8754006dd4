// Internal helpers shared by the host and device translation units of libmpcx.
#pragma once
#include <string>

void mpcx_set_error(const std::string& msg);
