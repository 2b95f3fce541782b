#include "core/core.hpp"
